#!/usr/bin/env python3
"""Two pipes (two contexts, two compute streams) fed from two threads against one pipe: what does overlapping two
independent jobs' kernels give on one device?  (GPU box)  usage: python tools/two_pipes_probe.py [batches=300]"""
import os, sys, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx0 = _lib.Context(0)
mp3 = bytes(ctx0.encode_pcm(synth_pcm(10000, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
msg = "The quick brown fox jumps over the lazy dog, again and again and.."
ref = bytes(ctx0.hide_message(mp3, msg)["data"])


def pump(pipe, k, out):
    sub = got = 0
    ok = True
    while got < k:
        while sub < k and pipe.submit([mp3], [msg]) is not None:
            sub += 1
        _t, res = pipe.collect()
        if got % 32 == 0:
            ok = ok and bytes(res[0]["data"]) == ref
        del res
        got += 1
    out.append(ok)


for n_pipes, depth, threads in ((1, 4, 3), (2, 4, 3), (2, 3, 2), (3, 3, 2))[:int(os.environ.get("PROBE_CASES", "4"))]:
    ctxs = [_lib.Context(0) for _ in range(n_pipes)]
    pipes = [_lib.Pipe(c, depth=depth, max_job_bytes=len(mp3) + 65536, scan_threads=threads) for c in ctxs]
    for p in pipes:
        pump(p, 12, [])
    oks = []
    ths = [threading.Thread(target=pump, args=(p, nb, oks)) for p in pipes]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    for p in pipes:
        p.close()
    print(json.dumps({"pipes": n_pipes, "depth": depth, "scan_threads": threads, "batches": nb * n_pipes, "ms_per_batch": round(dt / (nb * n_pipes) * 1e3, 4),
                      "frames_per_s": round(10000 * nb * n_pipes / dt, 1), "same_bytes": all(oks)}), flush=True)
    for c in ctxs:
        c.close()
