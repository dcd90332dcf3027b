#!/bin/bash
# GPU box: Huffman kernel time for each lanes-per-wave variant (MP3S_HUF_LANES) at two batch sizes.
cd "$GRAFT_REPO_ROOT"
for frames in 10000 60000; do
for l in 64 32 16 8; do
  MP3S_HUF_LANES=$l python bench.py --steps 10 --warmup 2 --frames $frames --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames=$frames lanes=$l', d['value'], d['kernels_ms_per_step']['k_dec_huffman'], d['parity_checked'])"
done; done
