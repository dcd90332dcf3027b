#!/usr/bin/env python3
"""Every second pipe instance is 25 % slower: which CPUs / NUMA nodes are involved?  (GPU box)"""
import os, sys, time, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
libc = ctypes.CDLL(None)
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
msg = os.environ.get("MSG", "The quick brown fox")
pin = os.environ.get("PIN")
if pin:
    os.sched_setaffinity(0, set(range(*map(int, pin.split("-")))))
for it in range(6):
    cpu0 = libc.sched_getcpu()
    pipe = _lib.Pipe(ctx, depth=4, max_job_bytes=len(mp3) + 65536, scan_threads=3)
    def run(k):
        sub = got = 0
        while got < k:
            while sub < k and pipe.submit([mp3], [msg]) is not None:
                sub += 1
            _t, res = pipe.collect(); del res; got += 1
    run(12)
    t0 = time.perf_counter(); run(150); dt = time.perf_counter() - t0
    st = pipe.stats()
    ctx.profile_select(None); ctx.profile_enable(True)
    run(40)
    pr = {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_collect().items()}
    ctx.profile_enable(False)
    pipe.close()
    print(json.dumps({"pipe": it, "kernels_ms": pr, "ms_per_batch": round(dt / 150 * 1e3, 4), "main_cpu_at_create": cpu0, "main_cpu_now": libc.sched_getcpu(), "last_span_ms": round(st["last_device_span_ms"], 3)}), flush=True)
