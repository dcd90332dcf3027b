#!/usr/bin/env python3
"""Where Steganography.hide_message(quiet) spends its time beyond the native call (GPU box): cProfile over 50 calls on a 10 000-frame file."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import Steganography, _lib
from synth_pcm import synth_pcm
ctx = _lib.default_context()
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
src, dst = os.path.join(td, "in.mp3"), os.path.join(td, "out.mp3")
open(src, "wb").write(mp3)
s = Steganography(quiet=True)
for _ in range(5):
    s.hide_message(src, dst, "x" * 64)
n = 50
t0 = time.perf_counter()
for _ in range(n):
    s.hide_message(src, dst, "x" * 64)
print("facade ms per file: %.3f" % ((time.perf_counter() - t0) / n * 1e3))
t0 = time.perf_counter()
for _ in range(n):
    ctx.hide_message(mp3, "x" * 64)
print("native call on bytes ms: %.3f" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    s.hide_message(src, dst, "x" * 64)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
