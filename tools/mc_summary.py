#!/usr/bin/env python3
"""rocprofv3 --memory-copy-trace of tools/repro_config5.py: the copies that take more than 50 us, in time order, from the start of the last mix's calls:
direction, stream, duration, gap to the copy before.  usage: python tools/mc_summary.py <rocprofv3 output dir> [tail_ms=60]"""
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
end = int(rows[-1]["End_Timestamp"])
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end - s > tail * 1e6 or e - s < 50_000:
        continue
    out.append({"t_ms": round((s - (end - tail * 1e6)) / 1e6, 3), "dir": r["Direction"].replace("MEMORY_COPY_", ""), "stream": r["Stream_Id"], "us": round((e - s) / 1e3, 1),
                "src": r["Source_Agent_Id"], "dst": r["Destination_Agent_Id"]})
json.dump(out, sys.stdout)
