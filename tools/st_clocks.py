"""Phase clocks of k_dec_stream (library built with -DMP3S_ST_CLOCKS=1: every wave writes the shader clocks it spent per phase into a device
array, read back through mp3s_debug_st_clocks).  usage (GPU box): python tools/st_clocks.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mp3stego import _lib
from synth_pcm import synth_pcm
n = 10000
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(n, seed=7), 44100, 128, None)["mp3"])
for _ in range(2):
    pcm = ctx.decode_stream(mp3, _lib.MP3S_PCM_I16)["pcm"]
import ctypes
buf = np.zeros(4096 * 8, dtype=np.uint32)
assert _lib.lib().mp3s_debug_st_clocks(buf.ctypes.data_as(ctypes.c_void_p)) == 0
d = buf.reshape(-1, 8).astype(np.int64)
d = d[d[:, 4] > 0]
names = ["setup (tables to LDS ...)", "requantise + alias + G", "IMDCT rows", "synthesis", "whole wave"]
for k in range(5):
    print("%-28s median %8d  max %8d clocks" % (names[k], np.median(d[:, k]), d[:, k].max()))
print("waves", len(d))
