#!/bin/bash
# Every soak one after the other on other cases than the default runs' (SOAK_SEED), SECONDS each, with a heartbeat (a silent run is taken to be hung
# after seven minutes).  usage (gpurun): SOAK_SEED=100000 bash tools/soak_all.sh TAG [seconds=150]   ->  gpurun_out/TAG_soak_*.txt
TAG=${1:-soak}; S=${2:-150}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( while true; do date +%s >> gpurun_out/hb.txt; sleep 45; done ) &
HB=$!
rc=0
for t in soak soak_mutant_hide soak_blocks soak_chains soak_select_long soak_rate_spectra soak_pipe; do
  SEED=$SOAK_SEED timeout -k 10 $((S + 240)) python tools/$t.py $S > gpurun_out/${TAG}_$t.txt 2> gpurun_out/${TAG}_$t.err; r=$?
  echo "$t rc=$r: $(tail -1 gpurun_out/${TAG}_$t.txt | cut -c1-300)"
  [ $r -ne 0 ] && rc=$r
done
kill $HB
exit $rc
