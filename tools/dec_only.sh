cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 500 python bench.py --no-cpu-baseline --no-config5 --steps 50 > gpurun_out/bx.out 2> gpurun_out/bx.err; python -c "
import json
d=json.loads(open('gpurun_out/bx.out').read().strip().splitlines()[-1])
do=d['decode_only']; print('exact', do['ms_per_step'], 'serial', do['serial_ms_per_step'], 'fast', do['float_fast']['ms_per_step'], 'i16', do['int16']['ms_per_step'], d['parity_checked'], d['value'])
"
