#!/usr/bin/env python3
"""config-5's mixed-block stream alone (GPU box): 250 synthesised frames x 40, decoded to int16 and float32 through the one-file call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
import frame_synth
ctx = _lib.Context(0)
for name, kw in (("mixed", dict(seed=104, block_types=(0, 2), allow_mixed=True, use_reservoir=True)), ("long", dict(seed=101, block_types=(0,), use_reservoir=True))):
    seed = kw.pop("seed")
    data = frame_synth.make_stream(seed, 250, **kw) * 40
    for fmt, fn in ((_lib.MP3S_PCM_I16, "int16"), (_lib.MP3S_PCM_F32, "float32")):
        for _ in range(3):
            ctx.decode_stream(data, fmt)
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.decode_stream(data, fmt)
        print(name, fn, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), ctx.run_stats())
