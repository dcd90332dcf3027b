#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (mean per dispatch).  usage: pmc_summary.py <dir> <tag>"""
import csv, glob, os, sys, collections, json
root, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, f"pmc_{tag}_*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", "")
        res[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k in sorted(res):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(res[k].items())}
    print(k)
    for c, v in out[k].items():
        print(f"   {c:28s} {v:16.1f}   (n={len(res[k][c])})")
json.dump(out, open(os.path.join(root, f"pmc_{tag}_summary.json"), "w"), indent=1)
