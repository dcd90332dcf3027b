#!/usr/bin/env python3
"""MP3S_TRACE=1 python tools/pipe_trace.py depth threads batches: per-job scan times and CPUs of the pipe's workers (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
depth, threads, batches = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
pipe = _lib.Pipe(ctx, depth=depth, max_job_bytes=len(mp3) + 65536, scan_threads=threads)
sub = got = 0
t0 = time.perf_counter()
while got < batches:
    while sub < batches and pipe.submit([mp3], ["The quick brown fox"]) is not None:
        sub += 1
    _t, res = pipe.collect(); del res; got += 1
print("ms per batch", (time.perf_counter() - t0) / batches * 1e3, pipe.stats())
pipe.close(); ctx.close()
