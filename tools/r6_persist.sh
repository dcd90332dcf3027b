#!/bin/bash
# the persistent rate loop: parity with every launch forced through it (MP3S_RATE_PERSIST=2), then A/B on one box
cd "$GRAFT_REPO_ROOT"
MP3S_RATE_PERSIST=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_select.py tests/test_pipe.py tests/test_batch_hide.py tests/test_sharded.py -m gpu -x -q > gpurun_out/t_persist.txt 2>&1; rc=$?; echo "persist-forced tests rc $rc"; tail -3 gpurun_out/t_persist.txt
[ $rc -eq 0 ] || exit 1
ROUNDS=2 bash tools/abx.sh o5@MP3S_RATE_PERSIST=0 o6@MP3S_RATE_PERSIST=0 o5 o6 > gpurun_out/ab_persist.txt 2>&1; grep -A2 "^==" gpurun_out/ab_persist.txt | grep -v "^--$"
