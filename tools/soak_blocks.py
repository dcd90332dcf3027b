"""One-off soak on the GPU box: mutated and damaged streams (tests/test_fuzz.py generators over tests/test.mp3 and the
decode corpus: reservoir, mixed blocks, MS, mono, CRC, ID3, false syncs, truncation) decoded as 2-7 blocks must give
what one call gives for the whole stream, which tools/soak.py pins to the oracle.
usage (via gpurun): python tools/soak_blocks.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib, sharded
import test_fuzz as tf

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = mlib.Context(0)
gd = os.path.join(ROOT, 'tests', 'golden')
data = open(os.path.join(gd, "test.mp3"), "rb").read()
g = np.load(os.path.join(gd, "g7_decode_corpus.npz"))
names = sorted({k.split("__")[0] for k in g.files})
t_end = time.time() + budget
stats = {"ok": 0, "both_reject": 0, "bad": 0}
seed = 9000 + int(os.environ.get("SOAK_SEED", "0"))
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    src = data if seed % 3 else g[names[seed % len(names)] + "__mp3"].tobytes()
    gen = tf.header_mutants(mlib, src, 10, seed) if seed % 2 else tf.mutants(src, 10, seed)
    for m in gen:
        if rng.integers(0, 4) == 0:
            m = m[:len(m) - int(rng.integers(1, 900))]
        try:
            whole = ctx.decode_stream(m, mlib.MP3S_PCM_F64)
        except mlib.Mp3sError:
            whole = None
        world = int(rng.integers(2, 8))
        comm = sharded.LocalComm(world)
        got, err = None, False
        try:
            for r in range(world):
                comm.rank = r
                got = sharded.decode_sharded(ctx, m, comm, mlib.MP3S_PCM_F64)
        except mlib.Mp3sError:
            err = True
        if whole is None:
            # a stream the whole-file decode rejects may still decode in blocks that do not reach the damage: not compared
            stats["both_reject"] += 1
            continue
        if err or got["pcm"].shape != whole["pcm"].shape or got["pcm"].tobytes() != whole["pcm"].tobytes() \
                or not np.array_equal(got["bits"], whole["bits"]):
            stats["bad"] += 1
            print("BLOCK mismatch seed", seed, "world", world, "err", err, flush=True)
        else:
            stats["ok"] += 1
print(stats)
