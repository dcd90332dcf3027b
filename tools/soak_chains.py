"""One-off soak on the GPU box of the encoder's serial chains (message cursor, inherited address / quantiser state):
random signals with silent stretches, random message lengths, as single files, as batches (mp3s_hide_messages) and as
blocks of one stream (mp3stego.sharded); every output against the oracle's decode -> int16 -> encode.
usage (via gpurun): python tools/soak_chains.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib, sharded
import oracle_lib as orc
from synth_pcm import synth_pcm

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = mlib.Context(0)
t_end = time.time() + budget
stats = {"single": 0, "batch_files": 0, "sharded": 0, "bad": 0}
rates = (32000, 44100, 48000)
kb = (32, 48, 64, 96, 128, 192, 256, 320)
seed = 5000 + int(os.environ.get("SOAK_SEED", "0"))


def signal(rng, n, rate):
    pcm = synth_pcm(n, seed=int(rng.integers(1, 1 << 30)), rate=rate)
    for _ in range(int(rng.integers(0, 4))):                         # silent stretches, sometimes one channel only
        a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, 8)))
        if rng.integers(0, 3) == 0:
            pcm[a * 1152:b * 1152, int(rng.integers(0, 2))] = 0
        else:
            pcm[a * 1152:b * 1152] = 0
    if rng.integers(0, 6) == 0:
        pcm = (pcm.astype(np.int32) // 64).astype(np.int16)          # quiet: few tables per granule
    return pcm


def message(rng):
    k = int(rng.integers(0, 7))
    if k == 0:
        return None
    ln = [0, 1, 6, 30, 130, 400, 2000][k]
    ln = int(rng.integers(max(ln // 2, 0), ln + 1))
    return "".join(chr(int(c)) for c in rng.integers(32, 127, size=ln))


def expect(mp3, msg):
    d = orc.decode(mp3)
    bits = None if msg is None else np.array(mlib.message_frame(msg))
    return orc.encode(orc.pcm_to_i16(d["pcm"]), int(d["sampling_rate"]), int(d["bit_rate"]) // 1000, bits)


while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    files, msgs = [], []
    for _ in range(int(rng.integers(1, 12))):
        rate = rates[int(rng.integers(0, 3))] if rng.integers(0, 3) == 0 else 44100
        kbps = kb[int(rng.integers(0, len(kb)))] if rng.integers(0, 3) == 0 else 128
        n = int(rng.integers(1, 70))
        files.append(ctx.encode_pcm(signal(rng, n, rate), rate, kbps, None)["mp3"])
        msgs.append(message(rng))
    want = [expect(f, m) for f, m in zip(files, msgs)]
    out = ctx.hide_messages(files, msgs)
    for i, (o, w) in enumerate(zip(out, want)):
        ok = not isinstance(o, Exception) and w["rc"] == 0 and o["data"] == w["mp3"] and o["hide_offset"] == w["hide_offset"] \
            and o["too_long"] == bool(w["too_long"])
        stats["batch_files"] += 1
        if not ok:
            stats["bad"] += 1
            print("BATCH mismatch seed", seed, "file", i, flush=True)
    i = int(rng.integers(0, len(files)))
    single = ctx.clear_file(files[i]) if msgs[i] is None else ctx.hide_message(files[i], msgs[i])
    stats["single"] += 1
    if single["data"] != want[i]["mp3"]:
        stats["bad"] += 1
        print("SINGLE mismatch seed", seed, "file", i, flush=True)
    world = int(rng.integers(2, 6))
    comm = sharded.LocalComm(world)
    res = None
    for r in range(world):
        comm.rank = r
        res = sharded.reencode_sharded(ctx, files[i], msgs[i], comm)
    stats["sharded"] += 1
    if res["data"] != want[i]["mp3"] or res["hide_offset"] != want[i]["hide_offset"]:
        stats["bad"] += 1
        print("SHARDED mismatch seed", seed, "file", i, "world", world, flush=True)
print(stats)
