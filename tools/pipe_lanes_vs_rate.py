"""Twelve fresh contexts, a public pipe (depth 4, one scan thread) on each: what its rehearsal chose and what it sustains on 10 000-frame jobs.  usage: python tools/pipe_lanes_vs_rate.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
c0 = _lib.Context(0)
mp3 = bytes(c0.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
for i in range(12):
    c = _lib.Context(0)
    pipe = _lib.Pipe(c, depth=4, max_job_bytes=8 << 20, scan_threads=1)
    n_jobs, inflight, done = 600, 0, 0
    t0 = None
    while done < n_jobs:
        while inflight < 4 and done + inflight < n_jobs and pipe.submit([mp3], ["x" * 64]) is not None:
            inflight += 1
        pipe.collect(); inflight -= 1; done += 1
        if done == 100:
            t0 = time.perf_counter()
    dt = time.perf_counter() - t0
    st = pipe.stats()
    print("ctx %2d: lanes 0x%x queue_shared %d rehearsal %.1f ms  %.3f M frames/s" % (i, st.get("lanes", -1), st.get("queue_shared", -1), st.get("rehearsal_us", 0) / 1e3, (n_jobs - 100) * 10000 / dt / 1e6), flush=True)
    pipe.close(); c.close()
