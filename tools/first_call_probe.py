import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
t0=time.perf_counter()
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx0 = _lib.Context(0)
pcm = synth_pcm(10000, seed=7)
mp3 = bytes(ctx0.encode_pcm(pcm, 44100, 128, None)["mp3"])
ctx0.close()
print("setup", time.perf_counter()-t0, file=sys.stderr)
t0=time.perf_counter(); c = _lib.Context(0); t1=time.perf_counter()
r = c.hide_message(mp3, "x"*64); t2=time.perf_counter()
r = c.hide_message(mp3, "x"*64); t3=time.perf_counter()
print("ctx create %.2f ms, first call %.2f ms, second %.2f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3), file=sys.stderr)
print(c.run_stats(), file=sys.stderr)
c2 = _lib.Context(0); t4=time.perf_counter(); r = c2.hide_message(mp3, "x"*64); t5=time.perf_counter()
print("second context first call %.2f ms" % ((t5-t4)*1e3), c2.run_stats(), file=sys.stderr)
