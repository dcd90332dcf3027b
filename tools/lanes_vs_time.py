"""Eight fresh contexts, one after the other: what each pipe's rehearsal chose (run_stats lanes) and what its one-file calls take.  usage: python tools/lanes_vs_time.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
c0 = _lib.Context(0)
mp3 = bytes(c0.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
for i in range(8):
    c = _lib.Context(0)
    t0 = time.perf_counter(); c.hide_message(mp3, "x" * 64); t_first = time.perf_counter() - t0
    for _ in range(3):
        c.hide_message(mp3, "x" * 64)
    t0 = time.perf_counter()
    for _ in range(30):
        c.hide_message(mp3, "x" * 64)
    t = (time.perf_counter() - t0) / 30
    rs = c.run_stats()
    print("ctx %d: lanes 0x%x rehearsal %.1f ms (%d)  first call %.1f ms  steady %.4f ms" % (i, rs["lanes"], rs["rehearsal_us"] / 1e3, rs["rehearsals"], t_first * 1e3, t * 1e3), flush=True)
    c.close()
