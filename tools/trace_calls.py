"""Six one-file calls in a row under MP3S_TRACE (stderr): where the host's time goes in steady state.  usage: MP3S_TRACE=1 python tools/trace_calls.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
c = _lib.Context(0)
mp3 = bytes(c.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
for i in range(6):
    print("=== call", i, file=sys.stderr, flush=True)
    t0 = time.perf_counter(); r = c.hide_message(mp3, "x" * 64); t1 = time.perf_counter()
    print("=== %.3f ms" % ((t1 - t0) * 1e3), file=sys.stderr, flush=True)
    del r
