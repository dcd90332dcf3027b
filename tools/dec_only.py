#!/usr/bin/env python3
"""The decode transforms alone on the bench's 10 000-frame batch (BASELINE configs[1] without the Huffman stage): ms per launch of
mp3s_decode_transform_dev to float64 / float32 / int16 with MP3S_OPT_FUSED_DECODE on (one wave-local stream kernel per format class) and off
(IMDCT and synthesis as two kernels with S in device memory; the float formats always take those), and that both give the same bytes.  usage (GPU box): python tools/dec_only.py [frames [format ...]]"""
import ctypes as C, hashlib, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mp3stego import _lib
from synth_pcm import synth_pcm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
only = sys.argv[2:]
ctx = _lib.Context(0)
L = _lib.lib()
mp3 = bytes(ctx.encode_pcm(synth_pcm(n, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
p = _lib.parse_stream(mp3)
d_is, d_si, d_hdr = ctx.to_device(p["is"]), ctx.to_device(p["si"]), ctx.to_device(p["hdr"])
out = {"frames": n}
for fmt, name, esz in ((_lib.MP3S_PCM_F64, "float64", 8), (_lib.MP3S_PCM_F32, "float32", 4), (_lib.MP3S_PCM_I16, "int16", 2)):
    if only and name not in only:
        continue
    d_pcm = ctx.alloc(n * 2304 * esz)
    row = {}
    for fused in (1, 0):
        ctx.set_option("fused_decode", fused)
        for _ in range(3):
            _lib.check(L.mp3s_decode_transform_dev(ctx.handle, d_is, d_si, d_hdr, n, 2, 0, fmt, d_pcm))
        ctx.sync()
        ctx.timer_start()
        for _ in range(30):
            _lib.check(L.mp3s_decode_transform_dev(ctx.handle, d_is, d_si, d_hdr, n, 2, 0, fmt, d_pcm))
        ms = ctx.timer_stop() / 30
        got = ctx.download(d_pcm, np.uint8, (n * 2304 * esz,))
        row["fused" if fused else "two_kernels"] = {"ms": round(ms, 4), "sha256": hashlib.sha256(got.tobytes()).hexdigest()[:16]}
    row["same_bytes"] = row["fused"]["sha256"] == row["two_kernels"]["sha256"]
    out[name] = row
    ctx.free(d_pcm)
ctx.set_option("fused_decode", 1)
print(json.dumps(out))
