#!/usr/bin/env python3
"""ms per one-file hide_message call for (first chunk, later chunks) plans on a 10 000- and a 100 000-frame file"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
fs = _lib.parse_stream(mp3)["frame_size"]
big = mp3[:int(fs[:9999].sum())] * 10
msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
def timed(data, n):
    r = ctx.hide_message(data, msg); del r
    t0 = time.perf_counter()
    for _ in range(n):
        r = ctx.hide_message(data, msg); del r
    return round((time.perf_counter() - t0) / n * 1e3, 3)
out = {"small": {}, "large": {}}
for first, rest in ((0, 0), (768, 16000), (1024, 16000), (1536, 16000), (2048, 16000), (1024, 4608), (1536, 4352), (1024, 3072), (1536, 2900), (2048, 4096), (10000, 16000)):
    ctx.set_option("first_chunk_frames", first); ctx.set_option("chunk_frames", rest)
    out["small"]["%d+%d" % (first, rest)] = timed(mp3, 30)
for first, rest in ((0, 0), (1536, 16000), (4096, 16000), (8192, 16000), (2048, 8192), (4096, 12288), (2048, 12288), (16000, 16000)):
    ctx.set_option("first_chunk_frames", first); ctx.set_option("chunk_frames", rest)
    out["large"]["%d+%d" % (first, rest)] = timed(big, 8)
print(json.dumps(out, indent=1))
