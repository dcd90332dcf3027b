#!/bin/bash
# sweep the number of decoding lanes per wave of k_dec_huffman (MP3S_HUF_ACTIVE) on the GPU box
for a in ${HUF_SWEEP:-8 16 32 64}; do
  MP3S_HUF_ACTIVE=$a timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('active', $a, d['kernels_ms_per_step']['k_dec_huffman'], d['parity_checked'])"
done
