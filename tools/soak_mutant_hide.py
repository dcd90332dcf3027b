"""One-off soak on the GPU box: hide_message / clear_file on mutated and damaged stereo streams, one by one and as a
batch, against the oracle's decode -> int16 -> encode of the same bytes (wild PCM, repeated last frames, frames the
host parser has to take over ...).   usage (via gpurun): python tools/soak_mutant_hide.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib
import oracle_lib as orc
import test_fuzz as tf

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = mlib.Context(0)
gd = os.path.join(ROOT, 'tests', 'golden')
data = open(os.path.join(gd, "test.mp3"), "rb").read()
g = np.load(os.path.join(gd, "g7_decode_corpus.npz"))
names = sorted({k.split("__")[0] for k in g.files})
import contextlib
import tempfile
sink_fd = os.open(os.path.join(tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None), "sink.mp3"), os.O_RDWR | os.O_CREAT, 0o600)


@contextlib.contextmanager
def mlib_options(c, chunk_frames):
    old = c.set_option("chunk_frames", chunk_frames)
    try:
        yield
    finally:
        c.set_option("chunk_frames", old)


t_end = time.time() + budget
stats = {"ok": 0, "both_reject": 0, "bad": 0}
seed = 30000 + int(os.environ.get("SOAK_SEED", "0"))
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    src = data if seed % 2 else g[names[seed % len(names)] + "__mp3"].tobytes()
    gen = tf.header_mutants(mlib, src, 8, seed) if seed % 3 == 0 else tf.mutants(src, 8, seed)
    files = []
    for m in gen:
        if rng.integers(0, 4) == 0:
            m = m[:len(m) - int(rng.integers(1, 900))]
        files.append(m)
    msgs = [None if rng.integers(0, 4) == 0 else "m" * int(rng.integers(0, 60)) for _ in files]
    out = ctx.hide_messages(files, msgs)
    for i, (f, msg, r) in enumerate(zip(files, msgs, out)):
        d = orc.decode(f)
        want = None
        if d["rc"] == 0 and d["n_frames"] > 0 and d["pcm"].ndim == 2 and d["pcm"].shape[1] == 2:
            bits = None if msg is None else np.array(mlib.message_frame(msg))
            e = orc.encode(orc.pcm_to_i16(d["pcm"]), int(d["sampling_rate"]), int(d["bit_rate"]) // 1000, bits)
            if e["rc"] == 0:
                want = e
        try:
            single = ctx.clear_file(f) if msg is None else ctx.hide_message(f, msg)
        except (mlib.Mp3sError, SystemExit):
            single = None
        # ... and the same call with its result written to a file descriptor (mp3s_*_fd: chunk by chunk), over something longer
        try:
            os.ftruncate(sink_fd, 0)                                   # (an empty file is written chunk by chunk, one that holds something at the end)
            if (seed + i) % 2:
                os.lseek(sink_fd, 0, os.SEEK_SET); os.write(sink_fd, b"\xaa" * (len(f) + 4096))
            with mlib_options(ctx, 64 if seed % 5 == 0 else 0):
                rf = ctx.recode_to_fd(f, msg, sink_fd)
            os.lseek(sink_fd, 0, os.SEEK_SET)
            to_fd = os.read(sink_fd, rf["len"] + 1)
        except (mlib.Mp3sError, SystemExit):
            to_fd = None
        if (single is None) != (to_fd is None) or (single is not None and bytes(single["data"]) != to_fd):
            stats["bad"] += 1
            print("FD MISMATCH seed", seed, "file", i, single is None, to_fd is None, flush=True)
            continue
        if want is None:
            if isinstance(r, Exception) and single is None:
                stats["both_reject"] += 1
            else:
                stats["bad"] += 1
                print("ACCEPTED what the oracle rejects: seed", seed, "file", i, flush=True)
            continue
        if isinstance(r, Exception) or single is None or r["data"] != want["mp3"] or single["data"] != want["mp3"]:
            stats["bad"] += 1
            print("MISMATCH seed", seed, "file", i, type(r).__name__, single is None, flush=True)
        else:
            stats["ok"] += 1
print(stats)
