#!/usr/bin/env python3
"""steady-state decode jobs through the pipe; PIN=a-b restricts the process to CPUs [a, b) first (GPU box)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pin = os.environ.get("PIN")
if pin:
    a, b = map(int, pin.split("-")); os.sched_setaffinity(0, set(range(a, b)))
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
for depth, threads in ((4, 3), (2, 1), (6, 3)):
    pipe = _lib.Pipe(ctx, depth=depth, max_job_bytes=len(mp3) + 65536, scan_threads=threads)
    def run(k):
        sub = got = 0
        while got < k:
            while sub < k and pipe.submit_decode([mp3]) is not None:
                sub += 1
            _t, res = pipe.collect(); del res; got += 1
    run(10)
    t0 = time.perf_counter(); run(80); dt = time.perf_counter() - t0
    st = pipe.stats(); pipe.close()
    print(json.dumps({"pin": pin, "depth": depth, "threads": threads, "ms_per_batch": round(dt / 80 * 1e3, 3), "gb_per_s_down": round(46.08e6 * 80 / dt / 1e9, 1), "span_ms": round(st["last_device_span_ms"], 3)}), flush=True)
for i in range(3):
    r = ctx.decode_stream(mp3, _lib.MP3S_PCM_I16); del r
t0 = time.perf_counter()
for i in range(10):
    r = ctx.decode_stream(mp3, _lib.MP3S_PCM_I16); del r
print("sync decode_stream ms", (time.perf_counter() - t0) / 10 * 1e3)
