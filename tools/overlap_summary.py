#!/usr/bin/env python3
"""How much of the copy time of a rocprofv3 --kernel-trace --memory-copy-trace run lies under kernels.
usage: overlap_summary.py <rocprof output dir> <out.json>"""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
kern, copies = [], []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", "").split("<")[0]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), int(r.get("Bytes", 0) or 0) if "Bytes" in r else 0))
kern.sort()
# merged kernel busy intervals
busy = []
for a, b, _ in kern:
    if busy and a <= busy[-1][1]:
        busy[-1][1] = max(busy[-1][1], b)
    else:
        busy.append([a, b])
def under(a, b):
    t = 0
    for x, y in busy:
        if y <= a:
            continue
        if x >= b:
            break
        t += min(b, y) - max(a, x)
    return t
res = {}
for a, b, d, nbytes in copies:
    if b - a <= 0:
        continue
    e = res.setdefault(d, {"copies": 0, "ns": 0, "ns_under_kernels": 0, "ns_large": 0, "ns_large_under_kernels": 0})
    u = under(a, b)
    e["copies"] += 1; e["ns"] += b - a; e["ns_under_kernels"] += u
    if b - a > 20000:                      # the per-batch transfers (megabytes), not the small status words
        e["ns_large"] += b - a; e["ns_large_under_kernels"] += u
for d, e in res.items():
    e["fraction_under_kernels"] = round(e["ns_under_kernels"] / max(e["ns"], 1), 4)
    e["fraction_of_large_copies_under_kernels"] = round(e["ns_large_under_kernels"] / max(e["ns_large"], 1), 4)
span = (max(b for _, b, _ in kern) - min(a for a, _, _ in kern)) if kern else 0
summary = {"kernels": len(kern), "kernel_busy_ns": sum(y - x for x, y in busy), "trace_span_ns": span,
           "gpu_busy_fraction_of_span": round(sum(y - x for x, y in busy) / max(span, 1), 4), "copies": res}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1))
