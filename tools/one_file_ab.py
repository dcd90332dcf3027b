"""One-file calls with the pipe's events as dispatch signals (default) and as records, on one box, taking turns.  usage: python tools/one_file_ab.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
c0 = _lib.Context(0)
mp3 = bytes(c0.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
ctxs = {}
for v in (3, 2, 1, 0):
    c = _lib.Context(0); c.set_option("pipe_signals", v); ctxs[v] = c
    for _ in range(3):
        c.hide_message(mp3, "x" * 64)
for rnd in range(2):
    for v in (3, 2, 1, 0):
        c = ctxs[v]
        t0 = time.perf_counter()
        for _ in range(30):
            r = c.hide_message(mp3, "x" * 64)
        print("pipe_signals %d: %.4f ms per call" % (v, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
