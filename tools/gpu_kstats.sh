#!/bin/bash
# rocprofv3 kernel statistics of the resident steps only.  usage: bash tools/gpu_kstats.sh <tag>
TAG=${1:-k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline --resident-only > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_$TAG.err; echo "rocprof exit=$?"
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do head -16 $f | cut -c1-170; cp $f gpurun_out/${TAG}_kernel_stats.csv; done
rm -rf gpurun_out/prof_$TAG
