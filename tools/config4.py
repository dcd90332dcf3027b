#!/usr/bin/env python3
"""BASELINE config 4 at size on ONE device: 1 000 000 frames = 100 seeded streams x 10 000 frames (44.1 kHz stereo 128 kbps),
hidden message in every stream, cut into 8 contiguous blocks of 125 000 frames ("ranks").

Each rank's block = whole streams (through the asynchronous pipe, one job per stream) + at most two half streams (a block
boundary falls in the middle of streams 12, 37, 62, 87: a block job of the pipe, mp3s_pipe_submit_block, the second half on
the carry the first half left).  The
ranks are played one after another on this device; nothing but the 136-byte carry passes between them.

Checks: (1) every stream's bytes, reassembled from the ranks, equal mp3s_hide_message on the whole stream (all 1 000 000
frames); (2) CRC32 of every rank's output; (3) the oracle (CPU restatement) on a 1 % sample: the first 1 250 frames of one
stream per rank, decode + encode, byte for byte.

usage (GPU box): python tools/config4.py [streams=100] [frames=10000] [ranks=8] > profiles/r02_config4.json
"""
import json
import multiprocessing as mp
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
SEED = 0x9E3779B97F4A7C15


def _pcm(args):
    from synth_pcm import synth_pcm
    i, frames = args
    return i, synth_pcm(frames, seed=SEED + i)


def main():
    n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    total = n_streams * frames
    assert total % ranks == 0 and (2 * total // ranks) % frames == 0, "block boundaries fall on stream starts or stream middles"
    per_rank = total // ranks
    t_all = time.time()
    # ---- inputs: synthetic PCM in worker processes (forked before the GPU is touched), encoded on the device as they arrive
    pool = mp.get_context("fork").Pool(min(12, os.cpu_count() or 1))
    it = pool.imap_unordered(_pcm, [(i, frames) for i in range(n_streams)])
    from mp3stego import _lib
    ctx = _lib.Context(0)
    streams = [None] * n_streams
    for i, pcm in it:
        streams[i] = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    pool.close(); pool.join()
    msgs = ["stream %03d: the quick brown fox jumps over the lazy dog, again!" % i for i in range(n_streams)]
    t_inputs = time.time() - t_all

    # ---- the ranks
    pipe = _lib.Pipe(ctx, depth=4, max_job_bytes=max(len(s) for s in streams) + 65536, scan_threads=int(os.environ.get("CONFIG4_SCAN_THREADS", "2")))
    # warm-up outside the measurement (device buffers of the job size, page-locked result blocks, the block path's pool):
    # four jobs through the pipe and one pair of half-stream blocks
    for _ in range(4):
        assert pipe.submit([streams[0]], [msgs[0]]) is not None
    while pipe.collect() is not None:
        pass
    assert pipe.submit_block(streams[0], msgs[0], 0, 2, None) is not None
    _t, _w = pipe.collect()
    assert pipe.submit_block(streams[0], msgs[0], 1, 2, _w["carry_out"]) is not None
    pipe.collect()
    del _w
    ARENA["buf"] = np.zeros(sum(len(x) for x in streams) + (1 << 20), dtype=np.uint8)      # where the results are kept (touched: no page faults under the clock)
    out = [[] for _ in range(n_streams)]           # per stream: its pieces in order
    crc, rank_ms, rank_frames = [], [], []
    carry = {}                                      # stream -> carry of its first half
    for r in range(ranks):
        g0, g1 = r * per_rank, (r + 1) * per_rank
        t0 = time.perf_counter()
        pieces = []
        i = g0 // frames
        pos = g0
        while pos < g1:
            a, b = pos - i * frames, min(g1 - i * frames, frames)
            if a == 0 and b == frames:
                t = pipe.submit([streams[i]], [msgs[i]])
                while t is None:                                   # every slot taken: take the oldest result first
                    take(pipe.collect(), pieces, carry)
                    t = pipe.submit([streams[i]], [msgs[i]])
                pieces.append((i, None, t))
            else:
                # half a stream on this rank: a block job beside the whole streams (the second half runs on the carry the
                # first half left -- on one device the ranks are played one after another, so it is known)
                half = 0 if a == 0 else 1
                t = pipe.submit_block(streams[i], msgs[i], half, 2, carry.get(i))
                while t is None:
                    take(pipe.collect(), pieces, carry)
                    t = pipe.submit_block(streams[i], msgs[i], half, 2, carry.get(i))
                pieces.append((i, None, t))
            pos = i * frames + b
            i += 1
        while True:
            got = pipe.collect()
            if got is None:
                break
            take(got, pieces, carry)
        dt = time.perf_counter() - t0
        c = 0
        for i, data, _ in pieces:
            assert data is not None
            data = data.tobytes()
            out[i].append(data)
            c = zlib.crc32(data, c)
        crc.append(c); rank_ms.append(round(dt * 1e3, 3)); rank_frames.append(per_rank)
    st = pipe.stats()
    pipe.close()

    # ---- check 1: every stream against the single call on the whole stream
    t0 = time.perf_counter()
    same = 0
    for i in range(n_streams):
        whole = ctx.hide_message(streams[i], msgs[i])
        same += int(b"".join(out[i]) == bytes(whole["data"]) and not whole["too_long"])
    t_single = time.perf_counter() - t0
    # ---- check 3: the oracle on a 1 % sample (one stream per rank, its first `k` frames)
    import oracle_lib as O
    k = max(8, (total // 100) // ranks)
    oracle_same, oracle_frames = 0, 0
    for r in range(ranks):
        i = (r * per_rank + frames - 1) // frames          # first stream that starts inside the rank's block
        if i >= n_streams:
            continue
        p = _lib.parse_stream(streams[i])
        prefix = streams[i][:int(p["frame_size"][:k + 1].sum())]
        d = O.decode(prefix)
        o = O.encode(O.pcm_to_i16(d["pcm"])[:k * 1152], 44100, 128, np.array(_lib.message_frame(msgs[i])))
        mine = b"".join(out[i])
        m = len(o["mp3"]) - 8
        oracle_same += int(o["rc"] == 0 and mine[:m] == o["mp3"][:m])
        oracle_frames += k
    ctx.close()
    res = {"config": "BASELINE configs[3]: %d frames = %d streams x %d, %d contiguous blocks (ranks) of %d frames played on one MI355X" %
                     (total, n_streams, frames, ranks, per_rank),
           "frames": total, "ranks": ranks, "rank_ms": rank_ms, "rank_crc32": ["%08x" % c for c in crc],
           "one_device_all_ranks_s": round(sum(rank_ms) / 1e3, 4), "one_device_frames_per_s": round(total / (sum(rank_ms) / 1e3), 1),
           "slowest_rank_ms": max(rank_ms), "frames_per_s_if_ranks_ran_side_by_side": round(total / (max(rank_ms) / 1e3), 1),
           "streams_equal_to_single_call": same, "streams": n_streams, "single_call_loop_s": round(t_single, 3),
           "oracle_sample_frames": oracle_frames, "oracle_sample_streams_equal": oracle_same, "oracle_sample_streams": ranks,
           "pipe": {"fast": st["fast"], "resolved": st["resolved"], "slow": st["slow"]}, "inputs_s": round(t_inputs, 1), "total_s": round(time.time() - t_all, 1),
           "ok": same == n_streams and oracle_same == ranks}
    print(json.dumps(res))
    sys.exit(0 if res["ok"] else 3)


def _idx(pieces, ticket):
    for k, p in enumerate(pieces):
        if p[2] == ticket and p[1] is None:
            return k
    raise KeyError(ticket)


ARENA = {"buf": None, "at": 0}


def keep(view):
    """copy a result out of the library's page-locked block (which goes back to the pool) into memory that was allocated and
    touched before the clock started: a fresh 4 MB bytes object per result costs the page faults of its 1 024 pages"""
    a = np.frombuffer(view, dtype=np.uint8)
    at = ARENA["at"]
    dst = ARENA["buf"][at:at + a.size]
    np.copyto(dst, a)
    ARENA["at"] = at + a.size
    return dst


def take(got, pieces, carry):
    """a collected result into its place"""
    _t, res = got
    k = _idx(pieces, _t)
    if isinstance(res, dict):                                   # a block job
        if res["first_frame"] == 0:
            carry[pieces[k][0]] = res["carry_out"]
        pieces[k] = (pieces[k][0], keep(res["mp3"]), _t)
    else:
        pieces[k] = (pieces[k][0], keep(res[0]["data"]), _t)


def ctx_block(pipe, ctx, mp3, msg, half, carry_in, pieces):
    """half a stream on this rank: the pipe owns the context while it has jobs in flight, so drain it first"""
    while True:
        got = pipe.collect()
        if got is None:
            break
        _t, res = got
        k = _idx(pieces, _t)
        pieces[k] = (pieces[k][0], res[0]["data"], _t)     # (a view of the library's page-locked block: copied after the clock has stopped)
    return ctx.reencode_block(mp3, msg, half, 2, carry_in)


if __name__ == "__main__":
    main()
