#!/usr/bin/env python3
"""ms per decode_stream call (one file, int16 and float32) on a synthesised 10 000-frame stream, for chunk plans"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
import frame_synth
ctx = _lib.Context(0)
data = frame_synth.make_stream(101, 250, block_types=(0,), use_reservoir=True) * 40
def med(fmt, n=15):
    r = ctx.decode_stream(data, fmt); del r
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = ctx.decode_stream(data, fmt); del r; ts.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(ts), 3)
out = {}
plans = [(0, 0), (2048, 16000), (2500, 2500), (3334, 3334), (5000, 5000), (1024, 3000)]
for first, rest in plans:
    try:
        ctx.set_option("first_chunk_frames", first); ctx.set_option("chunk_frames", rest)
    except Exception:
        pass
    for env in ("", "1"):
        if env: os.environ["MP3S_NO_FILE_UP"] = "1"
        else: os.environ.pop("MP3S_NO_FILE_UP", None)
        out["%d+%d%s" % (first, rest, " no_file_up" if env else "")] = (med(_lib.MP3S_PCM_I16), med(_lib.MP3S_PCM_F32))
print(out)
