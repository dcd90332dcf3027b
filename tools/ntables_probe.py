"""How often does a unit's table count depend on the message bits it hides?  (Decides whether re-running the units on the
cursors the first pass predicts converges in a pass or two.)  usage: python tools/ntables_probe.py"""
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
n = 10000
pcm = synth_pcm(n, seed=5)
rng = np.random.default_rng(1)
res = {}
base = ctx.encode_pcm(pcm, 44100, 128, None)
nt0 = np.array(base["gr"]["n_tables"]).reshape(-1).copy()
prev = None
for name in ("a", "b"):
    bits = rng.integers(0, 2, size=n * 12 + 7, dtype=np.uint8)
    r = ctx.encode_pcm(pcm, 44100, 128, bits)
    nt = np.array(r["gr"]["n_tables"]).reshape(-1).copy()
    res["differs_from_no_message_" + name] = int((nt != nt0).sum())
    if prev is not None:
        res["differs_a_vs_b"] = int((nt != prev).sum())
    prev = nt
res["units"] = int(nt0.size)
res["units_with_3_tables"] = int((nt0 == 3).sum())
res["units_with_0_tables"] = int((nt0 == 0).sum())
print(json.dumps(res))
