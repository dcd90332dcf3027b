#!/bin/bash
# kernel timeline of a few resident steps (GPU box): rocprofv3 --kernel-trace, the middle steps printed as start / end offsets in us.  usage: bash tools/timeline.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --resident-only "$@" > gpurun_out/tl.out 2> gpurun_out/tl.err
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", ""), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ev.sort()
# the last 3 rate loops delimit two whole steps
rl = [i for i, e in enumerate(ev) if e[2].startswith("k_rate_loop")]
a, b = rl[-4], rl[-2]
t0 = ev[a][1]
for s, e, n, q in ev[a:b + 1]:
    print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
PY
rm -rf gpurun_out/tl
