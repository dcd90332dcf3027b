#!/bin/bash
# A/B of two builds (build/ab/A.so, B.so) on one box with rocprofv3's per-kernel attribution (the bench's own table lumps the fix-up launch
# with the synthesis).  usage (through gpurun): bash tools/abk.sh [pattern of kernel names]
PAT=${1:-synth|fixup|imdct}
cd "$GRAFT_REPO_ROOT"
L=mp3-steganography-lib_amd/mp3stego/libmp3s_hip.so
cp $L /tmp/keep.so
for v in ${ABK_SET:-A B}; do
  cp mp3-steganography-lib_amd/build/ab/$v.so $L; echo "== $v"
  cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab$v -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --resident-only --no-overlap > /dev/null 2> /dev/null
  python3 - "$PAT" gpurun_out/prof_ab$v <<'PY'
import csv, glob, re, sys
for f in glob.glob(sys.argv[2] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[1], r["Name"]):
            print("%-28s calls %4s  avg %8.1f us  min %8.1f" % (r["Name"].split("(")[0].replace("void ", "").replace("mp3s::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  rm -rf gpurun_out/prof_ab$v
done
cp /tmp/keep.so $L
