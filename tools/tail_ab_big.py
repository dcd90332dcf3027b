"""One-file calls of 100 000 frames on fresh contexts with the own pipe's tail stream (default) and without (pipe_tail 0).  usage: python tools/tail_ab_big.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
c0 = _lib.Context(0)
big = bytes(c0.encode_pcm(synth_pcm(100000, seed=8), 44100, 128, None)["mp3"])
for i in range(8):
    v = i & 1
    c = _lib.Context(0)
    if v == 0:
        c.set_option("pipe_tail", 0)
    for _ in range(2):
        c.hide_message(big, "x" * 64)
    t0 = time.perf_counter()
    for _ in range(6):
        c.hide_message(big, "x" * 64)
    tb = (time.perf_counter() - t0) / 6
    rs = c.run_stats()
    print("pipe_tail %s: lanes 0x%x  100k %.3f ms" % ("0" if v == 0 else "default", rs["lanes"], tb * 1e3), flush=True)
    c.close()
