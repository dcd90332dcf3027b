#!/bin/bash
# round 6, first GPU call: the sparse-spectra parity test on this build and (premise) on the round-5 build, then the whole GPU suite and a bench line
cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -k sparse -q -x > gpurun_out/t_sparse_new.txt 2>&1; echo "new rc $?"
if [ -f mp3-steganography-lib_amd/build/ab/r5.so ]; then
  cp $L /tmp/keep.so; cp mp3-steganography-lib_amd/build/ab/r5.so $L
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -k sparse -q > gpurun_out/t_sparse_r5.txt 2>&1; echo "r5 rc $? (a failure is the premise)"
  cp /tmp/keep.so $L
fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.txt 2>&1; rc=$?; echo "gpu suite rc $rc"; tail -3 gpurun_out/t_gpu.txt
[ $rc -eq 0 ] && timeout -k 10 400 python bench.py > gpurun_out/bench_first.json 2> gpurun_out/bench_first.err; echo "bench rc $?"
