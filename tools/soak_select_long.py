"""One-off soak on the GPU box of the message-cursor selection over long reaches (several rounds of k_chain_select, the
state carried from round to round, stretches done again in short rounds): long random signals with silent stretches of
up to 40 frames, messages of up to 30 000 bits, as single streams (mp3s_encode_pcm) and as batches of files
(mp3s_hide_messages: one workgroup per stream, different reaches); every output against the oracle's encode.
usage (via gpurun): python tools/soak_select_long.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib
import oracle_lib as orc
from synth_pcm import synth_pcm

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = mlib.Context(0)
t_end = time.time() + budget
stats = {"single": 0, "first_pass_final": 0, "batch_files": 0, "bad": 0, "frames": 0}
seed = 9000 + int(os.environ.get("SOAK_SEED", "0"))


def signal(rng, n):
    pcm = synth_pcm(n, seed=int(rng.integers(1, 1 << 30)))
    for _ in range(int(rng.integers(0, 5))):
        a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, 41)))
        if rng.integers(0, 4) == 0:
            pcm[a * 1152:b * 1152, int(rng.integers(0, 2))] = 0
        else:
            pcm[a * 1152:b * 1152] = 0
    if rng.integers(0, 8) == 0:
        pcm = (pcm.astype(np.int32) // 64).astype(np.int16)
    return pcm


while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    n = int(rng.integers(260, 1600))
    pcm = signal(rng, n)
    nbits = int(rng.integers(1, 4)) * int(rng.integers(1, 10000))
    msg = rng.integers(0, 2, size=nbits).astype(np.uint8)
    r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
    stats["single"] += 1
    stats["frames"] += n
    stats["first_pass_final"] += int(r["rate_passes"] == 1)
    if o["rc"] != 0 or r["mp3"] != o["mp3"] or r["hide_offset"] != o["hide_offset"] or r["too_long"] != bool(o["too_long"]):
        stats["bad"] += 1
        print("SINGLE mismatch seed", seed, "frames", n, "bits", nbits, flush=True)
    if seed % 4 == 0:
        files, texts = [bytes(r["mp3"])], ["".join(chr(int(c)) for c in rng.integers(32, 127, size=int(rng.integers(1, 1500))))]
        for _ in range(int(rng.integers(1, 3))):
            m = int(rng.integers(20, 500))
            files.append(bytes(ctx.encode_pcm(signal(rng, m), 44100, 128, None)["mp3"]))
            texts.append("".join(chr(int(c)) for c in rng.integers(32, 127, size=int(rng.integers(1, 600)))))
        out = ctx.hide_messages(files, texts)
        for i, (f, t) in enumerate(zip(files, texts)):
            d = orc.decode(f)
            w = orc.encode(orc.pcm_to_i16(d["pcm"]), 44100, 128, np.array(mlib.message_frame(t), dtype=np.uint8))
            stats["batch_files"] += 1
            if isinstance(out[i], Exception) or w["rc"] != 0 or bytes(out[i]["data"]) != w["mp3"] or out[i]["hide_offset"] != w["hide_offset"]:
                stats["bad"] += 1
                print("BATCH mismatch seed", seed, "file", i, flush=True)
    if stats["single"] % 20 == 0:
        print(stats, flush=True)
print(stats)
