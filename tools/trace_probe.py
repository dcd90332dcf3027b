#!/usr/bin/env python3
"""MP3S_TRACE=1 python tools/trace_probe.py frames chunk_frames [repeat]: one traced hide_message call on a file of `frames` frames"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
if frames > 10000:
    fs = _lib.parse_stream(mp3)["frame_size"]
    mp3 = mp3[:int(fs[:9999].sum())] * (frames // 9999)
ctx.set_option("chunk_frames", chunk)
for i in range(3):
    ctx.hide_message(mp3, "hello")
sys.stderr.write("==== traced call\n")
t0 = time.perf_counter(); ctx.hide_message(mp3, "hello"); print((time.perf_counter() - t0) * 1e3, "ms")
