#!/usr/bin/env python3
"""One very large file through the one-file calls (GPU box): N frames (default 600 000: 250 MB of MP3, 2.8 GB of int16 PCM -- byte offsets beyond
2^31) as chunks through the context's own pipe against the same calls with MP3S_OPT_FILE_PIPELINE = 0 (the stages one after the other over the
whole file: one launch per kernel), for hide_message, clear_file and decode_stream to int16.  The stream is a 10 000-frame encode repeated (this
encoder uses no bit reservoir, so the repetition is a valid stream; the decoder's overlap and window history run across the seams).
usage (GPU box): python tools/big_file.py [frames=600000]"""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
ctx = _lib.Context(0)
base = bytes(ctx.encode_pcm(synth_pcm(10000, seed=4242), 44100, 128, None)["mp3"])
reps = (n + 9999) // 10000
big = base * reps
n = reps * 10000
msg = "a message for a very large file " * 40
out = {"frames": n, "mp3_bytes": len(big)}


def sha(b):
    return hashlib.sha256(b).hexdigest()[:16]


res = {}
for piped in (1, 0):
    ctx.set_option("file_pipeline", piped)
    row = {}
    for name, fn in (("hide_message", lambda: ctx.hide_message(big, msg)), ("clear_file", lambda: ctx.clear_file(big))):
        t0 = time.time()
        r = fn()
        row[name] = {"s": round(time.time() - t0, 3), "sha256": sha(bytes(r["data"])), "bytes": len(r["data"]), "n_frames": int(r["n_frames"]),
                     "too_long": bool(r["too_long"]), "hide_offset": int(r["hide_offset"])}
        del r
    t0 = time.time()
    d = ctx.decode_stream(big, _lib.MP3S_PCM_I16)
    pcm = np.ascontiguousarray(d["pcm"])
    row["decode_stream"] = {"s": round(time.time() - t0, 3), "sha256": sha(pcm.tobytes()), "pcm_bytes": int(pcm.nbytes), "n_frames": int(d["n_frames"]),
                            "bits": int(len(d["bits"]))}
    del d, pcm
    res["chunks" if piped else "whole_file"] = row
    print("...", "chunks" if piped else "whole file", json.dumps(row), file=sys.stderr, flush=True)
ctx.set_option("file_pipeline", 1)
rs = ctx.run_stats() if hasattr(ctx, "run_stats") else None
out.update(res)
same = all({k: v for k, v in res["chunks"][op].items() if k != "s"} == {k: v for k, v in res["whole_file"][op].items() if k != "s"} for op in res["chunks"])
out["same"] = bool(same) and res["chunks"]["decode_stream"]["n_frames"] == n and res["chunks"]["hide_message"]["n_frames"] == n
print(json.dumps(out))
sys.exit(0 if out["same"] else 1)
