#!/bin/bash
# PMC passes on the GPU box (via gpurun).  Counters only: no --stats / sys-trace with --pmc (pool rule).
# usage: bash tools/gpu_pmc.sh <tag> [program and arguments instead of the bench's resident step, e.g. python3 tools/dec_only.py]
TAG=${1:-r01}
shift
if [ $# -eq 0 ]; then set -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --resident-only; fi
PROG=("$@")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run_pass () {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- "${PROG[@]}" > gpurun_out/pmc_${TAG}_$name.out 2> gpurun_out/pmc_${TAG}_$name.err
  echo "pass $name exit=$?"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
ls gpurun_out/pmc_${TAG}_sq1/*/ 2>/dev/null | head
python3 tools/pmc_summary.py gpurun_out $TAG
