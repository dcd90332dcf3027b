#!/bin/bash
# two SQ counter passes over the resident bench steps, counters of ONE kernel (GPU box).  usage: bash tools/gpu_pmc_kernel.sh <tag> <kernel name> [bench args]
TAG=${1:-q}; KER=${2:-k_rate_loop}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run_pass () {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --resident-only --no-overlap $EXTRA > gpurun_out/pmc_${TAG}_$name.out 2> gpurun_out/pmc_${TAG}_$name.err
  echo "pass $name exit=$?"
}
EXTRA="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM
python3 tools/pmc_summary.py gpurun_out $TAG 2>/dev/null | grep -A40 "^per wave"
rm -rf gpurun_out/pmc_${TAG}_sq1 gpurun_out/pmc_${TAG}_sq2
