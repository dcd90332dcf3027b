#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, a rocprofv3 kernel trace of the resident steps, and a kernel + memory-copy
# trace of the asynchronous pipe in steady state (copies under kernels).  usage: bash tools/gpu_profile_r02.sh <tag>
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit=$?"
tail -c 1500 gpurun_out/bench_$TAG.json; tail -5 gpurun_out/bench_$TAG.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --resident-only > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_$TAG.err; echo "rocprof exit=$?"
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do head -14 $f; cp $f gpurun_out/${TAG}_kernel_stats.csv; done
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/pipe_$TAG -- python3 tools/pipe_trace.py 4 3 120 > gpurun_out/pipe_$TAG.out 2> gpurun_out/pipe_$TAG.err; echo "pipe trace exit=$?"
tail -2 gpurun_out/pipe_$TAG.out
python3 tools/overlap_summary.py gpurun_out/pipe_$TAG gpurun_out/${TAG}_pipe_overlap.json
rm -rf gpurun_out/pipe_$TAG gpurun_out/prof_$TAG
