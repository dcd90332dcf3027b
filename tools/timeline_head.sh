#!/bin/bash
# the FIRST and LAST kernels of the timed region of a short run (GPU box): what filling and draining the three streams costs.  usage: bash tools/timeline_head.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --resident-only --no-live-pmc "$@" > gpurun_out/tl.out 2> gpurun_out/tl.err
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", ""), r.get("Queue_Id", "?")) for r in rows]
ev.sort()
# the timed region: the last 20 rate loops and everything behind the rate loop in front of them
rl = [i for i, e in enumerate(ev) if e[2].startswith("k_rate_loop")]
start = rl[-21] + 1 if len(rl) > 20 else 0
while start < len(ev) and ev[start][0] < ev[rl[-21]][1] + 20000 and len(rl) > 20:   # (the tail of the step in front, the verdict's copies)
    start += 1
reg = ev[start:]
t0 = reg[0][0]
last_end = max(e[1] for e in reg if not e[2].startswith("__amd_rocclr") or e[3] != reg[0][3])
print("timed region: %d kernels, %.1f us from the first start to the end of the last tail" % (len(reg), (max(e[1] for e in reg[:-2]) - t0) / 1e3))
for s, e, n, q in reg[:24]:
    print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
print("   ...")
for s, e, n, q in reg[-14:]:
    print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
PY
rm -rf gpurun_out/tl
