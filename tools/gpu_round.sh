#!/bin/bash
# The whole measurement set of a round on the GPU box: GPU tests + bench line, rocprofv3 kernel trace + pipe overlap trace,
# PMC passes, config 4 at size.  usage: bash tools/gpu_round.sh <tag>
TAG=${1:-r02c}
bash tools/gpu_r02.sh $TAG
bash tools/gpu_profile_r02.sh $TAG
bash tools/gpu_pmc.sh $TAG
cd "$GRAFT_REPO_ROOT"
timeout 1500 python tools/config4.py > gpurun_out/${TAG}_config4.json 2> gpurun_out/${TAG}_config4.err; echo "config4 exit=$?"
tail -c 1200 gpurun_out/${TAG}_config4.json
