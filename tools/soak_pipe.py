#!/usr/bin/env python3
"""Soak of the asynchronous pipe (GPU box): random jobs -- hide / clear / decode, one to twelve files, clean, cut, mutated and
synthetic streams of every kind, messages of every length -- through pipes of random depth / worker count, every result
against the synchronous calls (which are pinned to the oracle elsewhere).  usage: python tools/soak_pipe.py [seconds=90]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
import frame_synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ctx = _lib.Context(0)
pool = []
for i, (rate, kbps, n) in enumerate([(44100, 128, 400), (44100, 128, 37), (48000, 192, 90), (32000, 64, 60), (44100, 320, 25), (44100, 128, 1200)]):
    pcm = synth_pcm(n, rate=rate, seed=100 + i)
    if i % 2:
        pcm[: (n // 3) * 1152] = 0
    pool.append(bytes(ctx.encode_pcm(pcm, rate, kbps, None)["mp3"]))
pool.append(open(os.path.join(ROOT, "tests", "golden", "test.mp3"), "rb").read())
pool += [frame_synth.make_stream(50 + k, 70, block_types=(0, 1, 2, 3), allow_mixed=bool(k & 1), mode=(0, 1, 3)[k % 3], mode_ext=2 if k % 3 == 1 else 0,
                                 sr_idx=k % 3, use_reservoir=True) for k in range(6)]

def variant(f):
    r = rng.random()
    if r < 0.6:
        return f
    if r < 0.75:
        return f[: int(rng.integers(max(1, len(f) // 2), len(f)))]
    b = bytearray(f)
    for _ in range(int(rng.integers(1, 4))):
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)

def message():
    r = rng.random()
    if r < 0.15:
        return None
    n = int(rng.choice([0, 1, 3, 20, 64, 200, 900]))
    return "".join(chr(int(c)) for c in rng.integers(32, 127, size=n))

def same(a, b):
    if isinstance(b, Exception):
        return isinstance(a, _lib.Mp3sError) and a.code == b.code
    if isinstance(a, Exception):
        return False
    return bytes(a["data"]) == bytes(b["data"]) and a["too_long"] == b["too_long"] and a["hide_offset"] == b["hide_offset"] and a["n_frames"] == b["n_frames"]

t0 = time.time()
jobs_done = files_done = bad = pipes = 0
stats = {"fast": 0, "resolved": 0, "slow": 0}
while time.time() - t0 < budget:
    depth, threads = int(rng.integers(1, 6)), int(rng.integers(1, 5))
    pipe = _lib.Pipe(ctx, depth=depth, max_job_bytes=int(rng.choice([1 << 16, 1 << 19, 1 << 21])), scan_threads=threads)
    pipes += 1
    jobs = []
    for _ in range(int(rng.integers(5, 40))):
        files = [variant(pool[int(rng.integers(len(pool)))]) for _ in range(int(rng.choice([1, 1, 1, 2, 5, 12])))]
        kind = "d" if rng.random() < 0.25 else "h"
        jobs.append((kind, files, None if (kind == "d" or rng.random() < 0.1) else [message() for _ in files]))
    got, nxt = [], 0
    while len(got) < len(jobs):
        while nxt < len(jobs):
            k, files, msgs = jobs[nxt]
            t = pipe.submit_decode(files) if k == "d" else pipe.submit(files, msgs)
            if t is None:
                break
            nxt += 1
        got.append(pipe.collect()[1])
    st = pipe.stats()
    pipe.close()
    stats["fast"] += st["fast"]; stats["slow"] += st["slow"]; stats["resolved"] += st["resolved"]
    for (k, files, msgs), res in zip(jobs, got):
        if k == "d":
            want = []
            for f in files:
                try:
                    want.append(ctx.decode_file(f))
                except _lib.Mp3sError as e:
                    want.append(e)
            ok = all((isinstance(w, Exception) and isinstance(r, _lib.Mp3sError) and r.code == w.code) or
                     (not isinstance(w, Exception) and not isinstance(r, Exception) and bytes(r["data"]) == bytes(w["data"]) and np.array_equal(r["bits"], w["bits"]))
                     for r, w in zip(res, want))
        else:
            want = ctx.hide_messages(files, [None] * len(files) if msgs is None else msgs)
            ok = len(res) == len(want) and all(same(r, w) for r, w in zip(res, want))
        if not ok:
            bad += 1
            print("MISMATCH", k, [len(f) for f in files], msgs and [m and len(m) for m in msgs], flush=True)
        jobs_done += 1; files_done += len(files)
print(json.dumps({"seconds": round(time.time() - t0, 1), "pipes": pipes, "jobs": jobs_done, "files": files_done, "mismatches": bad, **stats}))
sys.exit(1 if bad else 0)
