#!/bin/bash
# a state.s measurements in one call -- the bench line (default flags: what the driver runs with --steps 20 --warmup 5 is run too), rocprofv3 kernel statistics of
# the resident steps, the PMC passes, the kernels alone.  usage (gpurun): bash tools/measure_state.sh <tag>
TAG=${1:-r06a}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_flags.json 2> gpurun_out/${TAG}_bench_driver_flags.err; echo "bench (driver flags) rc $?"
bash tools/gpu_kstats.sh $TAG > gpurun_out/${TAG}_kstats.txt 2>&1; echo "kstats rc $?"
bash tools/gpu_pmc.sh $TAG > gpurun_out/${TAG}_pmc.txt 2>&1; echo "pmc rc $?"
bash tools/kb.sh --no-overlap > gpurun_out/${TAG}_kernels_alone.txt 2>&1; echo "alone rc $?"
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench.json", "gpurun_out/${TAG}_bench_driver_flags.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, {k: d.get(k) for k in ("value", "value_min", "value_max", "ms_per_step", "dominant_kernel_ms", "sustained_frames_per_s", "step_hbm_bytes", "facade_ms_per_file", "c_call_ms_per_file", "first_call_ms", "decode_only_float32_exact_ms", "decode_only_int16_ms", "parity_checked")}, d["roofline"]["frac"], d["roofline_alu"])
PY
