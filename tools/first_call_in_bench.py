"""The first hide_message of a fresh context in a process that looks like bench.py's when it makes that call: the main context and three helper
contexts alive, each with a resident 10 000-frame batch behind it.  MP3S_TRACE is read once (a static), so the whole process is traced: the
output on stderr is cut to the lines after the marker."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
n = 10000
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(n, seed=7), 44100, 128, None)["mp3"])
helpers = [_lib.Context(0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3)]
for h in [ctx] + helpers:
    h.hide_message(mp3, "y" * 64) if h is ctx else h.decode_stream(mp3)
print("=== marker", file=sys.stderr, flush=True)
t0 = time.perf_counter(); c = _lib.Context(0); t1 = time.perf_counter()
r = c.hide_message(mp3, "x" * 64); t2 = time.perf_counter()
r = c.hide_message(mp3, "x" * 64); t3 = time.perf_counter()
print("=== ctx create %.2f ms, first call %.2f ms, second %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), c.run_stats(), file=sys.stderr)
