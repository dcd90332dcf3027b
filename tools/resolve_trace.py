"""Phase times of the host chain resolve on the bench's long message (MP3S_TRACE=1).  usage: python tools/resolve_trace.py [bytes]"""
import os
import sys

os.environ["MP3S_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib  # noqa: E402
from synth_pcm import synth_pcm  # noqa: E402

nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=5), 44100, 128, None)["mp3"])
text = "".join(chr(32 + (i * 7) % 90) for i in range(nbytes))
for i in range(3):
    sys.stderr.write("---- call %d\n" % i)
    r = ctx.hide_message(mp3, text)
    del r
