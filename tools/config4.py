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

`--procs N` (round 4): the same job as N PROCESSES, one per rank (one per GPU; MP3STEGO_DEVICE=k puts them all on device k):
every rank builds only the streams its block touches, owns its own context and pipe, and runs its block on its own clock
between two barriers.  The rank that holds the SECOND half of a stream starts it on the guess "message hidden, nothing
inherited" and receives the real 136-byte carry from the rank in front over gloo (dist.isend / irecv, exactly
mp3stego/sharded.py's rule: run again only if the block looked at its carry and the guess was wrong).  Nothing else passes
between ranks.  Rank 0 prints the per-rank times, the aggregate frames/s (all frames / slowest rank) and the same CRCs the
one-process run prints.  With WORLD_SIZE unset the ranks are started from here (tools/launch_ranks.py).
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
    argv = [a for a in sys.argv[1:]]
    procs = 0
    if "--procs" in argv:
        k = argv.index("--procs")
        procs = int(argv[k + 1])
        del argv[k:k + 2]
    n_streams = int(argv[0]) if len(argv) > 0 else 100
    frames = int(argv[1]) if len(argv) > 1 else 10000
    ranks = int(argv[2]) if len(argv) > 2 else (procs or 8)
    if procs:
        assert procs == ranks, "--procs N = N ranks, one process each"
        if "WORLD_SIZE" not in os.environ:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import launch_ranks
            sys.exit(launch_ranks.launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], procs))
        return rank_main(n_streams, frames, ranks)
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
           "slowest_rank_ms": max(rank_ms), "extrapolated_not_measured_frames_per_s_if_ranks_ran_side_by_side": round(total / (max(rank_ms) / 1e3), 1),
           "streams_equal_to_single_call": same, "streams": n_streams, "single_call_loop_s": round(t_single, 3),
           "oracle_sample_frames": oracle_frames, "oracle_sample_streams_equal": oracle_same, "oracle_sample_streams": ranks,
           "pipe": {"fast": st["fast"], "resolved": st["resolved"], "slow": st["slow"]}, "inputs_s": round(t_inputs, 1), "total_s": round(time.time() - t_all, 1),
           "ok": same == n_streams and oracle_same == ranks}
    print(json.dumps(res))
    sys.exit(0 if res["ok"] else 3)


def rank_main(n_streams, frames, ranks):
    """one rank of `--procs N` (RANK / WORLD_SIZE / MASTER_* from the launcher)"""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == ranks
    total = n_streams * frames
    assert total % ranks == 0 and (2 * total // ranks) % frames == 0, "block boundaries fall on stream starts or stream middles"
    per_rank = total // ranks
    g0, g1 = rank * per_rank, (rank + 1) * per_rank
    t_all = time.time()
    # the block's pieces in stream order: (stream, kind) with kind "whole" | "first" (frames [0, n/2) of it) | "second"
    plan, pos, i = [], g0, g0 // frames
    while pos < g1:
        a, b = pos - i * frames, min(g1 - i * frames, frames)
        plan.append((i, "whole" if (a == 0 and b == frames) else ("first" if a == 0 else "second")))
        pos = i * frames + b
        i += 1
    mine = sorted({i for i, _ in plan})
    pool = mp.get_context("fork").Pool(max(1, min(len(mine), (os.cpu_count() or 8) // world, 12)))   # forked before torch / the GPU
    it = pool.imap_unordered(_pcm, [(i, frames) for i in mine])
    import torch
    import torch.distributed as dist
    sys.stdout.flush()
    keep_out = os.dup(1)
    os.dup2(2, 1)                                   # (the gloo transport announces its connections on stdout)
    try:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(keep_out, 1)
        os.close(keep_out)
    from mp3stego import _lib
    from mp3stego.sharded import CARRY_WORDS, _PAST_MESSAGE, _same_effect
    ctx = _lib.Context(int(os.environ.get("MP3STEGO_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    streams = {}
    for i, pcm in it:
        streams[i] = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    pool.close(); pool.join()
    msgs = {i: "stream %03d: the quick brown fox jumps over the lazy dog, again!" % i for i in mine}
    t_inputs = time.time() - t_all
    pipe = _lib.Pipe(ctx, depth=4, max_job_bytes=max(len(s) for s in streams.values()) + 65536,
                     scan_threads=int(os.environ.get("CONFIG4_SCAN_THREADS", "2")))
    w = mine[0]
    for _ in range(4):
        assert pipe.submit([streams[w]], [msgs[w]]) is not None
    while pipe.collect() is not None:
        pass
    assert pipe.submit_block(streams[w], msgs[w], 0, 2, None) is not None
    _t, _w = pipe.collect()
    assert pipe.submit_block(streams[w], msgs[w], 1, 2, _w["carry_out"]) is not None
    pipe.collect()
    del _w
    ARENA["buf"] = np.zeros(2 * sum(len(x) for x in streams.values()) + (1 << 20), dtype=np.uint8)
    guess = np.zeros(CARRY_WORDS, dtype=np.int64)
    guess[0] = _PAST_MESSAGE
    has_second = any(k == "second" for _, k in plan)
    has_first = any(k == "first" for _, k in plan)
    # the order of submission is free (results are put in place by piece): the first half at the block's END goes first, so that
    # its carry reaches the next rank early; then the second half at the block's start, on the guess; then the whole streams
    todo = sorted(range(len(plan)), key=lambda k: {"first": 0, "second": 1, "whole": 2}[plan[k][1]])
    data = [None] * len(plan)
    real_t = torch.zeros(CARRY_WORDS, dtype=torch.int64)
    dist.barrier()
    t0 = time.perf_counter()
    rreq = dist.irecv(real_t, rank - 1) if has_second else None
    sreq, reruns, carry_wait_ms = None, 0, 0.0
    flight = {}                                     # ticket -> (piece, carry it ran on)

    def submit(k, carry):
        i, kind = plan[k]
        if kind == "whole":
            t = pipe.submit([streams[i]], [msgs[i]])
        else:
            t = pipe.submit_block(streams[i], msgs[i], 0 if kind == "first" else 1, 2, carry)
        if t is not None:
            flight[t] = (k, carry)
        return t is not None

    queue = [(k, guess if plan[k][1] == "second" else None) for k in todo]
    while queue or flight:
        while queue and submit(*queue[0]):
            queue.pop(0)
        got = pipe.collect()
        if got is None:
            continue
        t, res = got
        k, carry = flight.pop(t)
        i, kind = plan[k]
        if kind == "whole":
            data[k] = keep(res[0]["data"])
            continue
        if kind == "first":
            sreq = dist.isend(torch.from_numpy(np.asarray(res["carry_out"], dtype=np.int64).copy()), rank + 1)
            data[k] = keep(res["mp3"])
            continue
        if carry is guess:                          # the second half came back from its run on the guess: what was the real carry?
            tw = time.perf_counter()
            rreq.wait()
            carry_wait_ms += (time.perf_counter() - tw) * 1e3
            real = real_t.numpy().copy()
            n_hide = len(_lib.message_frame(msgs[i]))
            live = min(int(real[0]), n_hide) < n_hide
            if not _same_effect(real, guess, n_hide) and (res["carry_used"] or live):
                reruns += 1
                queue.insert(0, (k, real))
                continue
        data[k] = keep(res["mp3"])
    dt = time.perf_counter() - t0
    if sreq is not None:
        sreq.wait()
    dist.barrier()
    t_job = time.perf_counter() - t0
    st = pipe.stats()
    pipe.close()
    crc = 0
    for d in data:
        crc = zlib.crc32(d.tobytes(), crc)
    # ---- check 1: every piece against the single call on its whole stream
    same, half_lens = 0, {}
    for k, (i, kind) in enumerate(plan):
        whole = ctx.hide_message(streams[i], msgs[i])
        wb, mine_b = bytes(whole["data"]), data[k].tobytes()
        if kind == "whole":
            ok = mine_b == wb
        elif kind == "first":
            ok = wb[:len(mine_b)] == mine_b
        else:
            ok = wb[len(wb) - len(mine_b):] == mine_b
        if kind != "whole":
            half_lens[(i, kind)] = (len(mine_b), len(wb))
        same += int(ok and not whole["too_long"])
    # ---- check 3: the oracle on a 1 % sample (the first stream that starts inside the block, its first k frames)
    import oracle_lib as O
    kf = max(8, (total // 100) // ranks)
    i = (rank * per_rank + frames - 1) // frames
    oracle_same, oracle_frames = 0, 0
    if i in streams:
        p = _lib.parse_stream(streams[i])
        prefix = streams[i][:int(p["frame_size"][:kf + 1].sum())]
        d = O.decode(prefix)
        o = O.encode(O.pcm_to_i16(d["pcm"])[:kf * 1152], 44100, 128, np.array(_lib.message_frame(msgs[i])))
        mine_b = data[[k for k, (j, kind) in enumerate(plan) if j == i][0]].tobytes()
        m = len(o["mp3"]) - 8
        oracle_same, oracle_frames = int(o["rc"] == 0 and mine_b[:m] == o["mp3"][:m]), kf
    ctx.close()
    part = {"rank": rank, "ms": round(dt * 1e3, 3), "crc": "%08x" % crc, "pieces": len(plan), "pieces_equal": same, "half_lens": half_lens,
            "oracle_same": oracle_same, "oracle_frames": oracle_frames, "reruns": reruns, "carry_wait_ms": round(carry_wait_ms, 3),
            "job_ms": round(t_job * 1e3, 3), "pipe": {"fast": st["fast"], "resolved": st["resolved"], "slow": st["slow"]},
            "inputs_s": round(t_inputs, 1), "device": ctx.device}
    parts = [None] * world if rank == 0 else None
    dist.gather_object(part, parts, dst=0)
    ok = True
    if rank == 0:
        halves = {}
        for pt in parts:
            halves.update(pt["half_lens"])
        halves_fit = all(halves[(i, "first")][0] + halves[(i, "second")][0] == halves[(i, "first")][1]
                         for (i, kind) in halves if kind == "first")
        rank_ms = [pt["ms"] for pt in parts]
        n_pieces = sum(pt["pieces"] for pt in parts)
        res = {"config": "BASELINE configs[3] as %d PROCESSES: %d frames = %d streams x %d, one contiguous block of %d frames per rank, "
                         "each rank with its own context and pipe, carries over gloo" % (world, total, n_streams, frames, per_rank),
               "frames": total, "procs": world, "devices": [pt["device"] for pt in parts], "rank_ms": rank_ms,
               "rank_crc32": [pt["crc"] for pt in parts], "slowest_rank_ms": max(rank_ms),
               "job_ms_between_barriers": max(pt["job_ms"] for pt in parts),
               "frames_per_s": round(total / (max(pt["job_ms"] for pt in parts) / 1e3), 1),
               "pieces": n_pieces, "pieces_equal_to_single_call": sum(pt["pieces_equal"] for pt in parts), "halves_fit": halves_fit,
               "oracle_sample_frames": sum(pt["oracle_frames"] for pt in parts), "oracle_sample_streams_equal": sum(pt["oracle_same"] for pt in parts),
               "oracle_sample_streams": sum(1 for pt in parts if pt["oracle_frames"]),
               "second_half_reruns": [pt["reruns"] for pt in parts], "carry_wait_ms": [pt["carry_wait_ms"] for pt in parts],
               "pipe": [pt["pipe"] for pt in parts], "inputs_s": max(pt["inputs_s"] for pt in parts), "total_s": round(time.time() - t_all, 1)}
        res["ok"] = ok = (res["pieces_equal_to_single_call"] == n_pieces and halves_fit and
                          res["oracle_sample_streams_equal"] == res["oracle_sample_streams"])
        print(json.dumps(res))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if flag[0] > 0.5 else 3)


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
