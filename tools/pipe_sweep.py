#!/usr/bin/env python3
"""Steady-state throughput of the asynchronous pipe over (jobs in flight, scan threads), on the bench's 10 000-frame stream.
usage: python tools/pipe_sweep.py [frames] [batches]   (GPU box)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib          # noqa: E402
from synth_pcm import synth_pcm    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(n, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
ref = bytes(ctx.hide_message(mp3, "The quick brown fox")["data"])
try:
    print(json.dumps({"affinity_cpus": len(os.sched_getaffinity(0)), "cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip()}))
except Exception as e:
    print(json.dumps({"env_probe": str(e)}))
for depth, threads in ((1, 1), (2, 1), (3, 2), (4, 2), (4, 3), (4, 4), (6, 4), (8, 6), (8, 8)):
    pipe = _lib.Pipe(ctx, depth=depth, max_job_bytes=len(mp3) + 65536, scan_threads=threads)

    def pump(k):
        sub = got = 0
        ok = True
        while got < k:
            while sub < k and pipe.submit([mp3], ["The quick brown fox"]) is not None:
                sub += 1
            _t, res = pipe.collect()
            if got % 32 == 0:
                ok = ok and bytes(res[0]["data"]) == ref
            del res
            got += 1
        return ok
    ok = pump(12)
    s0 = pipe.stats()
    t0 = time.perf_counter()
    ok = pump(batches) and ok
    dt = time.perf_counter() - t0
    s1 = pipe.stats()
    pipe.close()
    print(json.dumps({"depth": depth, "scan_threads": threads, "frames_per_s": round(n * batches / dt), "ms_per_batch": round(dt / batches * 1e3, 4),
                      "scan_ms": round((s1["scan_ms"] - s0["scan_ms"]) / batches, 3), "scan_cpu_ms": round((s1["scan_cpu_ms"] - s0["scan_cpu_ms"]) / batches, 3),
                      "issue_ms": round((s1["issue_ms"] - s0["issue_ms"]) / batches, 3), "fast": s1["fast"] - s0["fast"], "ok": ok}))
ctx.close()
