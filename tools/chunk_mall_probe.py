#!/usr/bin/env python3
"""a 100 000-frame file through mp3s_hide_message at ONE chunk size (frames; argv[1]), a few calls: the workload of the
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_chunk_mall.sh (does a transform group whose scratch outgrows the
256 MB Infinity Cache pay HBM where a smaller one pays MALL?) -- prints ms per call"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
chunk = int(sys.argv[1])
ctx = _lib.Context(0)
ctx.set_option("file_pipeline", 0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
fs = _lib.parse_stream(mp3)["frame_size"]
big = mp3[:int(fs[:9999].sum())] * 10
ctx.set_option("file_pipeline", 1)
ctx.set_option("chunk_frames", chunk); ctx.set_option("first_chunk_frames", chunk)
r = ctx.hide_message(big, "hello"); del r
t0 = time.perf_counter()
for _ in range(4):
    r = ctx.hide_message(big, "hello"); del r
print("chunk %d: %.3f ms per 99 990-frame file" % (chunk, (time.perf_counter() - t0) / 4 * 1e3))
