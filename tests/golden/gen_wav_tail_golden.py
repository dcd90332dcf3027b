#!/usr/bin/env python3
"""Golden for WAV files whose declared sample count is not a multiple of 1152 per channel (reference
encoder/MP3_Encoder.py:611-614, WAV_Reader.py:108): the reference encodes one more frame from whatever int16 values follow
in its buffer (np.fromfile was asked for twice the declared count) -- here a LIST chunk behind the data chunk -- and
raises IndexError when the file ends inside that frame.  Runs the upstream reference (build container only, refshim.py).

    python tests/golden/gen_wav_tail_golden.py        ->  tests/golden/g9_wav_tail.npz
"""
import io
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from synth_pcm import synth_pcm  # noqa: E402
from refshim import load_reference  # noqa: E402

load_reference()
from mp3stego.encoder.encoder import Encoder as REncoder  # noqa: E402


def wav_bytes(pcm, rate, trailer=b""):
    from scipy.io import wavfile
    f = io.BytesIO()
    wavfile.write(f, rate, pcm)
    return f.getvalue() + trailer


def main():
    pcm = synth_pcm(3, seed=99)[: 2 * 1152 + 500]                  # 2 frames + 500 samples per channel
    trailer = b"LIST" + struct.pack("<I", 4000) + bytes((i * 37 + 11) & 0xff for i in range(4000))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, data in (("tail", wav_bytes(pcm, 44100, trailer)), ("short", wav_bytes(pcm, 44100, trailer[:600]))):
            w, m = os.path.join(td, name + ".wav"), os.path.join(td, name + ".mp3")
            open(w, "wb").write(data)
            out[name + "_wav"] = np.frombuffer(data, dtype=np.uint8)
            try:
                REncoder(w, m, bitrate=128).encode(quiet=True)
                out[name + "_mp3"] = np.frombuffer(open(m, "rb").read(), dtype=np.uint8)
                out[name + "_error"] = np.array("")
            except Exception as e:                                  # noqa: BLE001
                out[name + "_mp3"] = np.zeros(0, dtype=np.uint8)
                out[name + "_error"] = np.array(type(e).__name__)
            print(name, len(out[name + "_mp3"]), str(out[name + "_error"]))
    np.savez_compressed(os.path.join(HERE, "g9_wav_tail.npz"), **out)


if __name__ == "__main__":
    main()
