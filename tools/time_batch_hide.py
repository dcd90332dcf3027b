"""One-off timing of the batch hide entry point: 250 cuts of 40 frames as one call vs the whole stream as one file
(run on the GPU box; MP3S_TRACE=1 prints the library's phase timings)."""
import sys, os, time
sys.path.insert(0, 'mp3-steganography-lib_amd'); sys.path.insert(0, 'tests')
import numpy as np
from mp3stego import _lib
from synth_pcm import synth_pcm
ctx = _lib.Context(0)
enc = ctx.encode_pcm(synth_pcm(10000), 44100, 128, None)["mp3"]
p = _lib.parse_stream(enc)
cuts = np.concatenate([[0], np.cumsum(p["frame_size"].astype(np.int64))])
shorts = [enc[int(cuts[a]):int(cuts[min(a + 40, 10000)])] for a in range(0, 10000, 40)]
notes = ["note %d" % i for i in range(len(shorts))]
for _ in range(3):
    t0 = time.time(); out = ctx.hide_messages(shorts, notes); print("batch", time.time() - t0)
t0 = time.time(); ctx.hide_message(enc, "note 1"); print("single 10k", time.time() - t0)
t0 = time.time(); ctx.hide_message(enc, "note 1"); print("single 10k", time.time() - t0)
