#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, then the bench line.  usage: bash tools/gpu_r02.sh <tag> [pytest args]
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit=$?"
tail -25 gpurun_out/pytest_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit=$?"
tail -c 6000 gpurun_out/bench_$TAG.json; tail -15 gpurun_out/bench_$TAG.err
