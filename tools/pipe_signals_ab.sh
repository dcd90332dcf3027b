cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 3 0 1 3; do
MP3S_PIPE_SIGNALS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-config5 --no-short-files --no-single-file-100k --sustained-seconds 2 --steps 50 > gpurun_out/sus.out 2> gpurun_out/sus.err
python -c "
import json
d=json.loads(open('gpurun_out/sus.out').read().strip().splitlines()[-1])
print('signals $v', 'sustained', round(d['sustained']['frames_per_s']/1e6,3), 'e2e', round(d['e2e_steady']['frames_per_s']/1e6,3), 'one file', d['regions']['single_file_10k']['ms_per_batch'], d['parity_checked'])
"
done
