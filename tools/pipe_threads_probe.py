#!/usr/bin/env python3
"""The pipe's steady state per scan-thread count and depth (10 000-frame jobs): ms per batch, host walk / issue per batch.
usage: python tools/pipe_threads_probe.py [batches]"""
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
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("MP3S_")}}
for parse in (1, 0):
    ctx.set_option("device_parse", parse)
    for depth, th in ((4, 1), (4, 2), (4, 3), (6, 2), (3, 1)):
        pipe = _lib.Pipe(ctx, depth=depth, max_job_bytes=len(mp3) + (1 << 16), scan_threads=th)
        sub = got = 0
        t0 = None
        while got < nb + 20:
            while sub < nb + 20 and pipe.submit([mp3], [msg]) is not None:
                sub += 1
            _t, res = pipe.collect()
            del res
            got += 1
            if got == 20:
                t0 = time.perf_counter(); st0 = pipe.stats()
        dt = time.perf_counter() - t0
        st = pipe.stats()
        pipe.close()
        out["parse%d_depth%d_threads%d" % (parse, depth, th)] = {"ms_per_batch": round(dt / nb * 1e3, 4), "host_ms": round((st["scan_ms"] - st0["scan_ms"]) / nb, 4),
                                                                 "issue_ms": round((st["issue_ms"] - st0["issue_ms"]) / nb, 4), "fast": st["fast"]}
print(json.dumps(out, indent=1))
