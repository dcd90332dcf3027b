#!/usr/bin/env python3
"""Soak of what the contexts of ONE process share (GPU box): the page-locked block pool, the choice of streams (lanes) a context's own pipe
makes, the constant tables, the per-thread error text.  T threads, each making and dropping contexts of its own on the same device, run a mix
of one-file calls (chunked through the context's own pipe: streams of several frame sizes, so pipes are outgrown and parked), decodes to every
format, multi-file calls and pipe jobs; every result against what ONE context gave for the same input before the threads started (those calls
are pinned to the oracle by tests/ and the other soaks).  INTEGRATION.md: calls on one context are serialised by the caller, contexts are
independent.  usage (GPU box): python tools/soak_threads.py [seconds=90] [threads=4]"""
import hashlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3stego import _lib
from synth_pcm import synth_pcm
import frame_synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed0 = int(os.environ.get("SOAK_SEED", "0"))

boot = _lib.Context(0)
pool = []
for i, (rate, kbps, n) in enumerate([(44100, 128, 3000), (44100, 64, 700), (48000, 192, 2500), (32000, 64, 60), (44100, 320, 900), (44100, 128, 37)]):
    pcm = synth_pcm(n, rate=rate, seed=seed0 + 700 + i)
    if i % 2:
        pcm[: (n // 3) * 1152] = 0
    pool.append(bytes(boot.encode_pcm(pcm, rate, kbps, None)["mp3"]))
pool.append(open(os.path.join(ROOT, "tests", "golden", "test.mp3"), "rb").read())
pool += [frame_synth.make_stream(seed0 + 50 + k, 70, block_types=(0, 1, 2, 3), allow_mixed=bool(k & 1), mode=(0, 1, 3)[k % 3], mode_ext=2 if k % 3 == 1 else 0,
                                 sr_idx=k % 3, use_reservoir=True) for k in range(4)]
pool.append(pool[0][:len(pool[0]) // 2 + 17])          # cut in a frame
pool.append(b"\x00" * 300)                              # refused
MSGS = [None, "", "a", "thread soak " * 5, "x" * 700]
FMTS = (_lib.MP3S_PCM_I16, _lib.MP3S_PCM_F32, _lib.MP3S_PCM_F64)


def digest(r):
    if isinstance(r, Exception):
        return ("err", getattr(r, "code", None))
    if "pcm" in r:
        return ("pcm", hashlib.sha256(np.ascontiguousarray(r["pcm"]).tobytes()).hexdigest(), r["n_frames"], bytes(np.asarray(r.get("bits", b""))[:64].tobytes()))
    return ("mp3", hashlib.sha256(bytes(r["data"])).hexdigest(), bool(r["too_long"]), int(r["hide_offset"]), int(r["n_frames"]))


def call(ctx, kind, i, m):
    try:
        if kind == "h":
            return ctx.clear_file(pool[i]) if MSGS[m] is None else ctx.hide_message(pool[i], MSGS[m])
        return ctx.decode_stream(pool[i], FMTS[m % 3])
    except _lib.Mp3sError as e:
        return e


want = {}
for i in range(len(pool)):
    for m in range(len(MSGS)):
        want[("h", i, m)] = digest(call(boot, "h", i, m))
    for m in range(3):
        want[("d", i, m)] = digest(call(boot, "d", i, m))
batch_files = [pool[5], pool[3], pool[7], pool[8], pool[-1], pool[5]]
batch_msgs = ["b%d" % k for k in range(len(batch_files))]
want_batch = [digest(r) for r in boot.hide_messages(batch_files, batch_msgs)]
boot.close()

errors, counts = [], [0] * T
t_end = time.time() + budget


def work(tid):
    rng = np.random.default_rng(seed0 + 9000 + tid)
    try:
        while time.time() < t_end:
            ctx = _lib.Context(0)
            if rng.random() < 0.3:
                ctx.set_option("chunk_frames", int(rng.choice([64, 300, 1000])))
            for phase in range(3):
                if phase == 1:
                    # the context belongs to a pipe while one exists: pipe jobs only, then the pipe goes
                    if rng.random() < 0.5:
                        pipe = _lib.Pipe(ctx, depth=int(rng.integers(1, 4)), max_job_bytes=1 << 21, scan_threads=int(rng.integers(1, 3)))
                        for _ in range(int(rng.integers(1, 6))):
                            assert pipe.submit(batch_files, batch_msgs) is not None
                            got = [digest(x) for x in pipe.collect()[1]]
                            if got != want_batch:
                                errors.append((tid, "pipe", str(got)[:200], str(want_batch)[:200]))
                            counts[tid] += 1
                        pipe.close()
                    continue
                for _ in range(int(rng.integers(2, 20))):
                    r = rng.random()
                    if r < 0.6:
                        key = ("h", int(rng.integers(len(pool))), int(rng.integers(len(MSGS))))
                        got = digest(call(ctx, *key))
                    elif r < 0.88:
                        key = ("d", int(rng.integers(len(pool))), int(rng.integers(3)))
                        got = digest(call(ctx, *key))
                    else:
                        key = "batch"
                        got = [digest(x) for x in ctx.hide_messages(batch_files, batch_msgs)]
                    exp = want_batch if key == "batch" else want[key]
                    if got != exp:
                        errors.append((tid, key, str(got)[:200], str(exp)[:200]))
                    counts[tid] += 1
                    if time.time() >= t_end:
                        break
            ctx.close()
    except Exception as e:                                       # noqa: BLE001
        import traceback
        errors.append((tid, "exception", repr(e), traceback.format_exc()[-600:]))


threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
t0 = time.time()
for t in threads:
    t.start()
while any(t.is_alive() for t in threads):
    time.sleep(20)
    print("... %d calls, %d errors" % (sum(counts), len(errors)), file=sys.stderr, flush=True)
for t in threads:
    t.join()
for e in errors[:10]:
    print("MISMATCH", e, flush=True)
print(json.dumps({"threads": T, "seconds": round(time.time() - t0, 1), "calls": sum(counts), "calls_by_thread": counts, "inputs": len(pool), "mismatches": len(errors)}))
sys.exit(1 if errors else 0)
