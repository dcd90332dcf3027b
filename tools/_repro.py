import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib
import oracle_lib as orc
import test_fuzz as tf
ctx = mlib.Context(0)
gd = os.path.join(ROOT, 'tests', 'golden')
data = open(os.path.join(gd, "test.mp3"), "rb").read()
g = np.load(os.path.join(gd, "g7_decode_corpus.npz"))
names = sorted({k.split("__")[0] for k in g.files})
for seed in (3891, 3897, 3929):
    src = data if seed % 3 else g[names[seed % len(names)] + "__mp3"].tobytes()
    gen = tf.header_mutants(mlib, src, 20, seed) if seed % 2 else tf.mutants(src, 20, seed)
    for i, m in enumerate(gen):
        o = orc.decode(m)
        try:
            r = ctx.decode_stream(m, mlib.MP3S_PCM_F64)
        except mlib.Mp3sError as e:
            if o["rc"] == 0:
                s = mlib.scan_stream(m)
                print(seed, i, "ERR", e, "| oracle frames", o["n_frames"], "gpu_ok", s["gpu_ok"], "nch", s["channels"], "hdr nch", sorted(set(s["hdr"]["nch"].tolist())), "sr", sorted(set(s["hdr"]["sr_idx"].tolist())))
