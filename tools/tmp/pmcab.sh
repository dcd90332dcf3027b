cd "$GRAFT_REPO_ROOT"
for v in A B; do
cp mp3-steganography-lib_amd/build/ab/$v.so mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
echo "== $v"; bash tools/gpu_pmc_kernel.sh ab$v k_rate_loop | grep -i "k_rate_loop"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_latest.json'))['k_rate_loop']
print({k:round(v/1e6,2) for k,v in d.items() if k.startswith('SQ_')})
PY
done
