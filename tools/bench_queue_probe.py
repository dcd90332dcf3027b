#!/usr/bin/env python3
"""one-file calls and the pipe in a process that holds as many contexts as bench.py does (hardware queue assignment)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0); aux = _lib.Context(0); aux2 = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
def timed(f, n):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f(); del r
    return round((time.perf_counter() - t0) / n * 1e3, 4)
out = {"ctx.hide_message": timed(lambda: ctx.hide_message(mp3, msg), 30), "aux.hide_message": timed(lambda: aux.hide_message(mp3, msg), 30)}
pctx = _lib.Context(0)
pipe = _lib.Pipe(pctx, depth=4, max_job_bytes=len(mp3) + (1 << 16), scan_threads=1)
sub = got = 0; nb = 300; t0 = None
while got < nb + 20:
    while sub < nb + 20 and pipe.submit([mp3], [msg]) is not None:
        sub += 1
    _t, res = pipe.collect(); del res; got += 1
    if got == 20: t0 = time.perf_counter()
out["pipe_ms_per_batch"] = round((time.perf_counter() - t0) / nb * 1e3, 4)
pipe.close()
out["ctx.hide_message again"] = timed(lambda: ctx.hide_message(mp3, msg), 30)
print(json.dumps(out, indent=1))
