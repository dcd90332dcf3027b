"""Phase clocks of k_dec_huffman (library built with -DMP3S_HUF_CLOCKS=1: the kernel writes shader-clock deltas into pairs 280..285 of every row;
the samples are wrong in such a build).  usage (GPU box): python tools/huf_clocks.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0)
n = 10000
src = _lib.Context(0)
mp3 = bytes(src.encode_pcm(synth_pcm(n, seed=7), 44100, 128, None)["mp3"])
sc = _lib.scan_stream(mp3)
L = _lib.lib()
d_blob, d_side = ctx.to_device(sc["blob"]), ctx.to_device(sc["side"])
d_is, d_si, d_st = ctx.alloc(n * 2304 * 2), ctx.alloc(n * 4 * 72), ctx.alloc(4 + 4 * n)
for _ in range(3):
    _lib.check(L.mp3s_huffman_decode_dev(ctx.handle, d_blob, d_side, n, 2, sc["max_part2_3_length"], d_is, d_si, d_st))
ctx.sync()
is_ = ctx.download(d_is, np.uint32, (n * 4, 288))
d = is_[:, 280:286].astype(np.int64)
names = ["tables to LDS", "staging + scalefactors", "symbol loop", "zero fill"]
for k in range(4):
    print("%-24s median %7d  max %7d clocks" % (names[k], np.median(d[:, k]), d[:, k].max()))
print("pairs walked: median", np.median(d[:, 4]), "max", d[:, 4].max())
