#!/usr/bin/env python3
"""Timeline of the LAST `n` kernels / copies of a rocprofv3 --kernel-trace --memory-copy-trace run, microseconds from the
first of them: what ran when, on which queue, and the gaps.  usage: timeline.py <rocprof output dir> [n_last] [marker kernel]"""
import csv, glob, os, sys
root = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ev = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), name[:40]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", r.get("Name", ""))[:40]))
ev.sort()
ev = ev[-n_last:]
t0 = ev[0][0]
last_end = {}
for a, b, q, name in ev:
    gap = (a - last_end[q]) / 1e3 if q in last_end else 0.0
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:7.1f} us  {q:6s} gap {gap:7.1f}  {name}")
    last_end[q] = b
