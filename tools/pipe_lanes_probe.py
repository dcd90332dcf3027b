#!/usr/bin/env python3
"""MP3S_TRACE=1: what the rehearsal decided for each new pipe and what the pipe then did (ms per 10 000-frame batch)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
nb = 200
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
ctxs = [_lib.Context(0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
for rep in range(2):
    for i, pctx in enumerate(ctxs):
        pipe = _lib.Pipe(pctx, depth=4, max_job_bytes=len(mp3) + (1 << 16), scan_threads=1)
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
        sys.stderr.write("RESULT ctx %d: %.4f ms per batch\n" % (i, dt / nb * 1e3))
