#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (mean per dispatch).  usage: pmc_summary.py <dir> <tag>"""
import csv, glob, os, sys, collections, json
root, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, f"pmc_{tag}_*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", "").split("<")[0]   # template args dropped
        res[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k in sorted(res):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(res[k].items())}
    print(k)
    for c, v in out[k].items():
        print(f"   {c:28s} {v:16.1f}   (n={len(res[k][c])})")
json.dump(out, open(os.path.join(root, f"pmc_{tag}_summary.json"), "w"), indent=1)
# what bench.py's roofline_alu reads: the per-launch counters of our kernels on the 10 000-frame batch
latest_pmc = {k: v for k, v in out.items() if k.startswith("k_")}
latest_pmc["frames"] = 10000
latest_pmc["source"] = f"profiles/{tag}_pmc_summary.json"
json.dump(latest_pmc, open(os.path.join(root, "pmc_latest.json"), "w"), indent=1)

# HBM bytes per launch the way MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE count 64-byte... units of
# 1 KiB in rocprofv3's derived form here (values are KiB), FETCH_SIZE under-reports by 2x on gfx950.
traffic = {}
for k, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and k.startswith("k_"):
        traffic[k] = {"fetch_bytes_raw": c["FETCH_SIZE"] * 1024, "fetch_bytes_x2": c["FETCH_SIZE"] * 2048,
                      "write_bytes": c["WRITE_SIZE"] * 1024,
                      "hbm_bytes_per_launch": c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024}
json.dump(traffic, open(os.path.join(root, f"pmc_{tag}_traffic.json"), "w"), indent=1)
latest = {k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}
latest["frames"] = 10000
json.dump(latest, open(os.path.join(root, "traffic_latest.json"), "w"), indent=1)

# one line per kernel, per wave (the shape the kernels are tuned by)
print("\nper wave:")
for k, v in out.items():
    w = v.get("SQ_WAVES", 0)
    if not w or "SQ_INSTS_SALU" not in v:
        continue
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print("%-20s waves %6d | VALU %6.0f SALU %5.0f SMEM %4.0f LDS %5.0f VMEM %4.0f | wave cycles %7.0f, waiting %2.0f%% (at a waitcnt %2.0f%%), "
          "LDS conflict cycles / LDS instr %.2f | VALU per launch %.1f M" %
          (k[:20], w, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_SMEM"] / w, v["SQ_INSTS_LDS"] / w, v["SQ_INSTS_VMEM"] / w, wc / w,
           100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v["SQ_INSTS_LDS"], 1),
           v["SQ_INSTS_VALU"] / 1e6))
