#!/bin/bash
# The GPU test suite once per non-default setting of the library's options (the environment provides a context's defaults: mp3s_api.cpp), to see
# that the alternative paths kept as options still give the reference's bytes.  Tests that assert a default's own behaviour (pass counts, kernel
# names) may fail under a setting for that reason: read the failures.  usage (gpurun): bash tools/option_sweep.sh  ->  gpurun_out/sweep_*.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in MP3S_FUSED_DECODE=0 MP3S_FUSED_ENCODE=1 MP3S_FILE_PIPELINE=0 MP3S_DEVICE_PARSE=0 MP3S_NO_SELECT=1 MP3S_NO_REDO=1 MP3S_FAST_IMDCT=0 "MP3S_RATE_SIGNALS=1 MP3S_PIPE_SIGNALS=3" MP3S_PIPE_DEC=1 MP3S_PIPE_TAIL=0 MP3S_NO_FILE_UP=1; do
  tag=$(echo "$s" | tr ' =' '__')
  env $s timeout -k 10 500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/sweep_$tag.txt 2>&1
  echo "$s: $(tail -1 gpurun_out/sweep_$tag.txt)"
  grep -E "^FAILED|^ERROR" gpurun_out/sweep_$tag.txt | cut -c1-200
done
