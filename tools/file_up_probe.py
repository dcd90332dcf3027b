#!/usr/bin/env python3
"""A/B inside one process: one-file calls with the whole file uploaded by the helper thread (default) and with MP3S_OPT_FILE_UP = 0
(each chunk's bytes by the calling thread), alternating, medians.  usage: file_up_probe.py [frames ...]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0)
base = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
fs = _lib.parse_stream(base)["frame_size"]
for frames in [int(a) for a in sys.argv[1:]] or [10000, 100000]:
    mp3 = base if frames <= 10000 else base[:int(fs[:9999].sum())] * (frames // 9999)
    t = {"file_up": [], "per_chunk": []}
    for i in range(4):
        ctx.hide_message(mp3, "hello")
    for rep in range(30 if frames <= 10000 else 10):
        for mode in ("file_up", "per_chunk"):
            ctx.set_option("file_up", 0 if mode == "per_chunk" else 1)
            t0 = time.perf_counter(); ctx.hide_message(mp3, "hello"); t[mode].append((time.perf_counter() - t0) * 1e3)
    ctx.set_option("file_up", 1)
    print(frames, "frames:", {k: (round(statistics.median(v), 3), round(min(v), 3)) for k, v in t.items()}, "(median, min) ms")
