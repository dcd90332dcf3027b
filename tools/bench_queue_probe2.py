#!/usr/bin/env python3
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0); aux = _lib.Context(0); aux2 = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
for i in range(4):
    t0 = time.perf_counter(); r = ctx.hide_message(mp3, msg); del r
    print((time.perf_counter() - t0) * 1e3)
