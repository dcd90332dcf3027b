#!/bin/bash
# HBM bytes per frame of the transform kernels against the chunk size of a one-file call: separate
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) per chunk size, summarised by tools/chunk_mall_summary.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in 4096 8192 12288 16382; do
  python3 tools/chunk_mall_probe.py $c > gpurun_out/mall_time_$c.txt 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/mall_${c}_$ctr -- python3 tools/chunk_mall_probe.py $c > gpurun_out/mall_${c}_$ctr.out 2> gpurun_out/mall_${c}_$ctr.err
    echo "chunk $c $ctr exit=$?"
  done
done
python3 tools/chunk_mall_summary.py gpurun_out > gpurun_out/r03_chunk_mall.json
cat gpurun_out/r03_chunk_mall.json
