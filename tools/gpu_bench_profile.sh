#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, then a rocprofv3 kernel trace of the same command.
# usage: bash tools/gpu_bench_profile.sh <tag>
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit=$?"
tail -c 3500 gpurun_out/bench_$TAG.json; tail -5 gpurun_out/bench_$TAG.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-short-files > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_$TAG.err; echo "rocprof exit=$?"
find gpurun_out/prof_$TAG -name "*stats*" | head
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do head -12 $f; done
