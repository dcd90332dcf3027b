"""How often does the first pass's cursor guess hold?  Random short messages into the bench stream and into cuts of it,
through the pipe; fast = the guess held, resolved = the host resolved the chains.  usage: python tools/guess_rate_probe.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib  # noqa: E402
from synth_pcm import synth_pcm  # noqa: E402

ctx = _lib.Context(0)
rng = np.random.default_rng(3)
out = {}
for seed in (5, 11):
    mp3 = bytes(ctx.encode_pcm(synth_pcm(2000, seed=seed), 44100, 128, None)["mp3"])
    for lo, hi in ((1, 8), (8, 40), (40, 200)):
        pipe = _lib.Pipe(ctx, depth=4, max_job_bytes=len(mp3) + 65536, scan_threads=2)
        jobs, sub, got = 120, 0, 0
        msgs = ["".join(chr(int(c)) for c in rng.integers(32, 127, size=int(rng.integers(lo, hi)))) for _ in range(jobs)]
        while got < jobs:
            while sub < jobs and pipe.submit([mp3], [msgs[sub]]) is not None:
                sub += 1
            pipe.collect()
            got += 1
        st = pipe.stats()
        pipe.close()
        out["seed%d_msg%d-%dB" % (seed, lo, hi)] = {"fast": st["fast"], "resolved": st["resolved"], "slow": st["slow"]}
print(json.dumps(out, indent=1))
