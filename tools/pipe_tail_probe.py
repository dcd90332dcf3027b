#!/usr/bin/env python3
"""the pipe with and without a stream of its own for a job's tail (selection, chain check, bit packing): ms per 10 000-frame batch"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
out = {}
for tail in (0, 2, 0, 2, 1):
    pctx = _lib.Context(0)
    pctx.set_option("pipe_tail", tail)
    for depth in (4, 6):
        pipe = _lib.Pipe(pctx, depth=depth, max_job_bytes=len(mp3) + (1 << 16), scan_threads=2)
        sub = got = 0
        t0 = None
        while got < nb + 20:
            while sub < nb + 20 and pipe.submit([mp3], [msg]) is not None:
                sub += 1
            _t, res = pipe.collect(); del res
            got += 1
            if got == 20:
                t0 = time.perf_counter()
        dt = time.perf_counter() - t0
        pipe.close()
        out.setdefault("tail%d_depth%d" % (tail, depth), []).append(round(dt / nb * 1e3, 4))
    pctx.close()
print(json.dumps(out))
