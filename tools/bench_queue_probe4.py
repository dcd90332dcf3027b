#!/usr/bin/env python3
"""is a context's resident kernel sequence slowed by streams created after it?"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib
from synth_pcm import synth_pcm
L = _lib.lib()
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = [_lib.Context(0) for _ in range(nctx)]
n = 2500
pcm = synth_pcm(n, seed=7)
mp3 = bytes(c[-1].encode_pcm(pcm, 44100, 128, None)["mp3"])
parsed = _lib.parse_stream(mp3)
msg = "The quick brown fox"
def resident(ctx, reps=50):
    d_is, d_si, d_hdr = ctx.to_device(parsed["is"]), ctx.to_device(parsed["si"]), ctx.to_device(parsed["hdr"])
    d_pcm, d_mdct = ctx.alloc(n * 2304 * 2), ctx.alloc(n * 2304 * 4)
    def step():
        _lib.check(L.mp3s_decode_transform_dev(ctx.handle, d_is, d_si, d_hdr, n, 2, 0, _lib.MP3S_PCM_I16, d_pcm))
        _lib.check(L.mp3s_encode_transform_dev(ctx.handle, d_pcm, d_hdr, n, d_mdct))
    for _ in range(5): step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps * 1e3
    for p in (d_is, d_si, d_hdr, d_pcm, d_mdct): ctx.free(p)
    return round(dt, 4)
out = {"contexts": nctx}
out["resident_before_any_pipe"] = [resident(x) for x in c]
r = c[-1].hide_message(mp3, msg); del r          # lanes are made now
out["resident_after_lanes"] = [resident(x) for x in c]
out["hide_message_ms"] = []
for x in c:
    x.hide_message(mp3, msg)
    t0 = time.perf_counter()
    for _ in range(20):
        r = x.hide_message(mp3, msg); del r
    out["hide_message_ms"].append(round((time.perf_counter() - t0) / 20 * 1e3, 3))
out["resident_after_own_pipes"] = [resident(x) for x in c]
print(json.dumps(out))
