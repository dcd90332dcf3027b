"""One-file calls (10 000 and 100 000 frames) on fresh contexts with the own pipe's tail stream rehearsed (default) and without one (pipe_tail 0).  usage: python tools/tail_ab.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
c0 = _lib.Context(0)
mp3 = bytes(c0.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
for i in range(16):
    v = i & 1
    c = _lib.Context(0)
    if v == 0:
        c.set_option("pipe_tail", 0)
    t0 = time.perf_counter(); c.hide_message(mp3, "x" * 64); t_first = time.perf_counter() - t0
    for _ in range(3):
        c.hide_message(mp3, "x" * 64)
    t0 = time.perf_counter()
    for _ in range(30):
        c.hide_message(mp3, "x" * 64)
    t = (time.perf_counter() - t0) / 30
    tb = 0.0
    rs = c.run_stats()
    print("pipe_tail %s: lanes 0x%x first call %.1f ms  10k %.4f ms  100k %.3f ms" % ("0" if v == 0 else "default", rs["lanes"], t_first * 1e3, t * 1e3, tb * 1e3), flush=True)
    c.close()
