#!/usr/bin/env python3
"""per chunk size: HBM bytes per frame (2 x FETCH_SIZE + WRITE_SIZE, KiB units as rocprofv3 reports them on gfx950: see
tools/pmc_summary.py) of the kernels whose scratch round trips could spill the Infinity Cache, and the time per file"""
import csv, glob, json, os, re, sys, collections
root = sys.argv[1]
out = {}
for c in (4096, 8192, 12288, 16382):
    per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(root, "mall_%d_%s" % (c, ctr), "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mp3s::", "").split("<")[0]
                if not k.startswith("k_") or row["Counter_Name"] != ctr:
                    continue
                per[k][ctr] += float(row["Counter_Value"])
                if ctr == "FETCH_SIZE":
                    per[k]["launches"] += 1
    frames = 99990 * 5 + 10000          # five calls on the file + the encode of its 10 000-frame source (stages one after the other)
    row = {"ms_per_file": None, "hbm_bytes_per_frame": {}}
    try:
        row["ms_per_file"] = float(re.search(r"([\d.]+) ms per", open(os.path.join(root, "mall_time_%d.txt" % c)).read()).group(1))
    except Exception:
        pass
    tot = 0.0
    for k in ("k_dec_imdct", "k_dec_synth_fast", "k_enc_analysis", "k_enc_mdct", "k_rate_loop", "k_dec_huffman", "k_enc_pack", "k_dec_parse"):
        if k in per:
            b = (2 * per[k]["FETCH_SIZE"] + per[k]["WRITE_SIZE"]) * 1024 / frames
            row["hbm_bytes_per_frame"][k] = round(b)
            tot += b
    row["hbm_bytes_per_frame"]["all_listed"] = round(tot)
    out["chunk_%d" % c] = row
out["note"] = ("scratch of a transform group: S = 18 432 bytes per frame (302 MB at 16 382 frames, 151 MB at 8 192: the Infinity Cache holds 256 MB), SB 9 216, mdct 9 216; "
               "algorithmic bytes per frame 14 208; counters are sums over five one-file calls on a 99 990-frame file plus one 10 000-frame encode")
print(json.dumps(out, indent=1))
