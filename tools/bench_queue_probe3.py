#!/usr/bin/env python3
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
mode = sys.argv[1]
msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
def timed(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f(); del r
    return round((time.perf_counter() - t0) / n * 1e3, 3)
c = [_lib.Context(0) for _ in range(3)]
pcm = synth_pcm(10000, seed=7)
out = {"mode": mode}
if mode == "enc_on_0":
    mp3 = bytes(c[0].encode_pcm(pcm, 44100, 128, None)["mp3"])
elif mode == "enc_on_1":
    mp3 = bytes(c[1].encode_pcm(pcm, 44100, 128, None)["mp3"])
elif mode == "enc_on_2":
    mp3 = bytes(c[2].encode_pcm(pcm, 44100, 128, None)["mp3"])
for order in ((0, 1, 2), (2, 1, 0)):
    for i in order:
        out["ctx%d_%s" % (i, "a" if order[0] == 0 else "b")] = timed(lambda: c[i].hide_message(mp3, msg))
print(json.dumps(out))
