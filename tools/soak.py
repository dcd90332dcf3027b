"""One-off soak on the GPU box: many more mutants / signals / settings than the committed tests; prints a summary.
usage (via gpurun): python tools/soak.py [seconds]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib
import oracle_lib as orc
import test_fuzz as tf
from synth_pcm import synth_pcm
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = mlib.Context(0)
gd = os.path.join(ROOT, 'tests', 'golden')
data = open(os.path.join(gd, "test.mp3"), "rb").read()
g = np.load(os.path.join(gd, "g7_decode_corpus.npz"))
names = sorted({k.split("__")[0] for k in g.files})
t_end = time.time() + budget
stats = {"dec_ok": 0, "dec_err": 0, "dec_bad": 0, "enc_ok": 0, "enc_err": 0, "enc_bad": 0}
seed = 1000 + int(os.environ.get("SOAK_SEED", "0"))     # (SOAK_SEED: other cases than the default run's)
seed0 = seed
rates = (32000, 44100, 48000); kb = (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    # ---- decoder mutants
    src = data if seed % 3 else g[names[seed % len(names)] + "__mp3"].tobytes()
    gen = tf.header_mutants(mlib, src, 20, seed) if seed % 2 else tf.mutants(src, 20, seed)
    for m in gen:
        o = orc.decode(m)
        try:
            r = ctx.decode_stream(m, mlib.MP3S_PCM_F64)
        except mlib.Mp3sError:
            stats["dec_err" if o["rc"] != 0 else "dec_bad"] += 1
            if o["rc"] == 0: print("DEC: lib error, oracle ok, seed", seed)
            continue
        same = o["rc"] == 0 and r["pcm"].shape == o["pcm"].shape and np.array_equal(r["pcm"].view(np.uint64), np.ascontiguousarray(o["pcm"]).view(np.uint64)) and np.array_equal(r["bits"], o["bits"])
        stats["dec_ok" if same else "dec_bad"] += 1
        if not same: print("DEC mismatch seed", seed)
    # ---- encoder: random signal shape / rate / bitrate / message
    n = int(rng.integers(2, 40)) * 1152
    kind = seed % 5
    if kind == 0: pcm = rng.integers(-32768, 32768, size=(n, 2))
    elif kind == 1: pcm = (rng.standard_normal((n, 2)) * rng.choice([3, 300, 8000])).clip(-32768, 32767)
    elif kind == 2: pcm = synth_pcm(n // 1152, seed=seed).astype(np.int64) * int(rng.integers(0, 4))
    elif kind == 3:
        pcm = np.zeros((n, 2)); k = int(rng.integers(0, n)); pcm[k:k + int(rng.integers(1, 2000))] = rng.integers(-32768, 32768)
    else: pcm = np.cumsum(rng.integers(-500, 501, size=(n, 2)), axis=0)
    pcm = np.ascontiguousarray(np.clip(pcm, -32768, 32767), dtype=np.int16)
    rate, kbps = rates[seed % 3], kb[(seed // 3) % 14]
    msg = rng.integers(0, 2, size=int(rng.choice([0, 5, 97, 1500]))).astype(np.uint8)
    o = orc.encode(pcm, rate, kbps, msg if len(msg) else None)
    try:
        r = ctx.encode_pcm(pcm, rate, kbps, msg if len(msg) else None)
    except mlib.Mp3sError as e:
        stats["enc_err" if o["rc"] != 0 else "enc_bad"] += 1
        if o["rc"] == 0: print("ENC: lib error, oracle ok, seed", seed, e)
        continue
    same = o["rc"] == 0 and r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"]
    stats["enc_ok" if same else "enc_bad"] += 1
    if not same: print("ENC mismatch seed", seed, rate, kbps, kind, len(msg))
print(stats, "seeds", seed - seed0, "from", seed0)
