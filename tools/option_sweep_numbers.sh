#!/bin/bash
# As option_sweep.sh, for the options that are numbers: one-file calls as chunks of 4 / 37 / 300 frames (thousands of chunk boundaries: halo, carried chain
# state, inherited scalefactors), first chunks of 4 / 5 frames, 16 / 62 decoding lanes in the Huffman kernel, 4 scan threads, the tail stream shared.
# (MP3S_CHUNK_FRAMES=4 takes 6.5 minutes: two gpurun calls.)  usage (gpurun): bash tools/option_sweep_numbers.sh  ->  gpurun_out/sweep2_*.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in MP3S_CHUNK_FRAMES=4 MP3S_CHUNK_FRAMES=37 "MP3S_CHUNK_FRAMES=300 MP3S_FIRST_CHUNK_FRAMES=5" MP3S_FIRST_CHUNK_FRAMES=4 MP3S_HUF_LANES=16 MP3S_HUF_LANES=62 MP3S_SCAN_THREADS=4 MP3S_PIPE_TAIL=2 MP3S_NO_NUMA=1; do
  tag=$(echo "$s" | tr ' =' '__')
  env $s timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/sweep2_$tag.txt 2>&1
  echo "$s: $(tail -1 gpurun_out/sweep2_$tag.txt)"
  grep -E "^FAILED|^ERROR" gpurun_out/sweep2_$tag.txt | cut -c1-200
done
