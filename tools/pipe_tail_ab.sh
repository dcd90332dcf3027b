#!/bin/bash
# the public pipe (bench's sustained / e2e regions) with MP3S_PIPE_TAIL 0 / 1, taking turns, several processes each (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 0 1 0 1 0 1; do
MP3S_PIPE_TAIL=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-config5 --no-short-files --no-single-file-100k --sustained-seconds 2 --steps 50 > gpurun_out/sus.out 2> gpurun_out/sus.err
python -c "
import json
d=json.loads(open('gpurun_out/sus.out').read().strip().splitlines()[-1])
print('pipe_tail $v', 'sustained', round(d['sustained']['frames_per_s']/1e6,3), 'e2e', round(d['e2e_steady']['frames_per_s']/1e6,3), d['parity_checked'])
"
done
