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
# regions = runs of kernels separated by idle gaps > 0.3 ms; the timed region is the one with 20 rate loops
regions, cur = [], [ev[0]]
for a, b in zip(ev, ev[1:]):
    if b[0] - max(e[1] for e in cur[-40:]) > 150000:
        regions.append(cur); cur = []
    cur.append(b)
regions.append(cur)
print("regions (rate loops in each):", [sum(1 for e in reg if e[2].startswith("k_rate_loop")) for reg in regions])
for reg in regions:
    n_rl = sum(1 for e in reg if e[2].startswith("k_rate_loop"))
    if n_rl != 20:
        continue
    t0 = reg[0][0]
    print("timed region: %d kernels, %.1f us from first start to last end" % (len(reg), (max(e[1] for e in reg) - t0) / 1e3))
    for s, e, n, q in reg[:26]:
        print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
    print("   ...")
    for s, e, n, q in reg[-14:]:
        print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
    break
PY
rm -rf gpurun_out/tl
