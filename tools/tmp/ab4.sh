cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
cp $L /tmp/keep.so
for r in 1 2 3 4; do for v in A B; do cp mp3-steganography-lib_amd/build/ab/$v.so $L; echo -n "$v "; bash tools/kb.sh --steps 300 | head -1; done; done
cp /tmp/keep.so $L
