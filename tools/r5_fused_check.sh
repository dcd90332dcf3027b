#!/bin/bash
# round 5: the fused decode kernel on the GPU box -- the decode tests first (a hang ends the call), then the resident steps with and
# without it, kernels alone and in the step.  usage (through gpurun): bash tools/r5_fused_check.sh [pytest -k expression]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
K=${1:-"fast or parity or corpus or dropin"}
timeout -k 10 420 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/r5_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r5_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stop"; exit $rc; fi
for f in 0 1; do
  echo "== MP3S_FUSED_DECODE=$f alone"; MP3S_FUSED_DECODE=$f bash tools/kb.sh --no-overlap || exit 1
  echo "== MP3S_FUSED_DECODE=$f step";  MP3S_FUSED_DECODE=$f bash tools/kb.sh || exit 1
done
