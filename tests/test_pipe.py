"""The asynchronous host-fed pipeline (include/mp3s.h section vii) and the on-device chain check it relies on
(mp3s_chain_resolve_dev): results are byte for byte those of the synchronous calls, which are pinned to the oracle
elsewhere (tests/test_batch_hide.py, tests/test_gpu_dropin.py) -- and against the oracle directly here."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expect(orc, mlib, mp3, message):
    d = orc.decode(mp3)
    pcm = orc.pcm_to_i16(d["pcm"])
    bits = None if message is None else np.array(mlib.message_frame(message))
    return orc.encode(pcm, int(d["sampling_rate"]), int(d["bit_rate"]) // 1000, bits)


def _drain(pipe, jobs):
    """keep the pipe full, collect in order; -> results per job"""
    out, nxt = [], 0
    while len(out) < len(jobs):
        while nxt < len(jobs):
            t = pipe.submit(*jobs[nxt])
            if t is None:
                break
            assert t == nxt
            nxt += 1
        t, res = pipe.collect()
        assert t == len(out)
        out.append(res)
    assert pipe.collect() is None
    return out


def test_pipe_matches_the_synchronous_calls_and_the_oracle(mlib, orc, golden_dir):
    from synth_pcm import synth_pcm
    ctx = mlib.Context(0)
    try:
        # the inputs: streams of several sizes and rates, one with silence (inherited addresses, E7)
        streams = []
        for i, (rate, kbps, n) in enumerate([(44100, 128, 300), (44100, 128, 41), (48000, 192, 60), (32000, 64, 25), (44100, 128, 120)]):
            pcm = synth_pcm(n, rate=rate, seed=77 + i)
            if i == 4:
                pcm[: 30 * 1152] = 0
                pcm[60 * 1152:80 * 1152] = 0
            streams.append(bytes(ctx.encode_pcm(pcm, rate, kbps, None)["mp3"]))
        g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
        names = sorted({k.split("__")[0] for k in g.files})
        corpus = [g[n + "__mp3"].tobytes() for n in names]           # mono, MS, reservoir, mixed blocks, CRC ...
        test_mp3 = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
        long_msg = "long " * 60                                       # > 1024 bits: the first pass cannot guess the cursors
        jobs = [
            ([streams[0]], ["a short message"]),
            ([streams[1]], None),                                     # clear
            ([streams[0], streams[1], streams[4]], ["one", None, "three"]),   # one device batch of three streams
            ([streams[2]], ["x"]),
            ([streams[0], streams[2]], ["two groups", "in one job"]), # two (rate, bitrate) groups: the synchronous path
            ([streams[4]], [long_msg]),                               # a silent start and more bits than the stream holds
            ([test_mp3], ["ddd"]),
            ([streams[3]], [""]),
            (corpus, ["m%d" % i for i in range(len(corpus))]),        # per-file status
            ([streams[0][:-700]], ["cut"]),                           # truncated last frame
            ([b"garbage" * 50], ["nope"]),
        ]
        want = []
        for files, msgs in jobs:
            want.append(ctx.hide_messages(files, [None] * len(files) if msgs is None else msgs))
        for depth, threads in ((3, 2), (1, 1), (5, 4)):
            pipe = mlib.Pipe(ctx, depth=depth, max_job_bytes=1 << 20, scan_threads=threads)
            try:
                got = _drain(pipe, jobs * 2)
                st = pipe.stats()
            finally:
                pipe.close()
            for k, res in enumerate(got):
                w = want[k % len(jobs)]
                assert len(res) == len(w)
                for a, b in zip(res, w):
                    if isinstance(b, Exception):
                        assert isinstance(a, mlib.Mp3sError) and a.code == b.code, (k, a, b)
                    else:
                        assert not isinstance(a, Exception), (k, a)
                        assert bytes(a["data"]) == bytes(b["data"]), k
                        assert a["too_long"] == b["too_long"] and a["hide_offset"] == b["hide_offset"], k
                        assert (a["kbps"], a["sampling_rate"], a["n_frames"]) == (b["kbps"], b["sampling_rate"], b["n_frames"]), k
            assert st["collected"] == 2 * len(jobs) and st["fast"] >= 2 * 5, st      # the plain jobs took the overlapped stages
            assert st["slow"] >= 2 * 2, st
            assert st["fast"] + st["resolved"] + st["slow"] == st["collected"], st
        # without the selection on the device (the first pass as it was until r02b) the long message's verdict is "redo":
        # its chains are resolved at collect time, on the job's own buffers
        keep = ctx.set_option("select", 0)
        try:
            pipe = mlib.Pipe(ctx, depth=3, max_job_bytes=1 << 20, scan_threads=2)
            try:
                got = _drain(pipe, [jobs[5], jobs[0], jobs[5]])
                st = pipe.stats()
            finally:
                pipe.close()
        finally:
            ctx.set_option("select", keep)
        assert st["resolved"] >= 2 and st["collected"] == 3, st
        for res, k in zip(got, (5, 0, 5)):
            assert bytes(res[0]["data"]) == bytes(want[k][0]["data"]) and res[0]["hide_offset"] == want[k][0]["hide_offset"], k
        # ... and the oracle on the jobs the device took alone
        for k in (0, 1, 3, 6, 7):
            files, msgs = jobs[k]
            o = _expect(orc, mlib, files[0], None if msgs is None else msgs[0])
            assert o["rc"] == 0 and bytes(want[k][0]["data"]) == o["mp3"], k
        for i in range(3):
            o = _expect(orc, mlib, jobs[2][0][i], jobs[2][1][i])
            assert bytes(want[2][i]["data"]) == o["mp3"], i
    finally:
        ctx.close()

def test_a_file_that_starts_in_silence_is_final_after_the_first_pass(mlib, orc):
    """the plan's reach follows the tables the input stream itself offers (silent units offer none): three seconds of
    silence in front of the music, a message that only starts behind them, several rounds of the selection; silences
    behind which units inherit addresses"""
    from synth_pcm import synth_pcm
    ctx = mlib.Context(0)
    try:
        pcm = synth_pcm(700, seed=91)
        pcm[: 115 * 1152] = 0
        # ... and silences further on: the first quiet granules behind them read addresses inherited across the silence
        # (E7), which the first pass cannot know; the device runs those few units again itself (mp3s_chain_redo_dev)
        pcm[300 * 1152:340 * 1152] = 0
        pcm[500 * 1152:503 * 1152, 1] = 0
        f = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
        msgs = ["a short one", "m" * 400, "n" * 700]
        pipe = mlib.Pipe(ctx, depth=2, max_job_bytes=1 << 20, scan_threads=1)
        try:
            got = _drain(pipe, [([f], [m]) for m in msgs])
            st = pipe.stats()
        finally:
            pipe.close()
        assert st["fast"] == 3 and st["resolved"] == 0 and st["slow"] == 0, st
        for res, m in zip(got, msgs):
            o = _expect(orc, mlib, f, m)
            assert o["rc"] == 0 and bytes(res[0]["data"]) == o["mp3"] and res[0]["hide_offset"] == o["hide_offset"], m
    finally:
        ctx.close()


def test_pipe_staging_too_small_takes_the_other_path(mlib):
    from synth_pcm import synth_pcm
    ctx = mlib.Context(0)
    try:
        mp3 = bytes(ctx.encode_pcm(synth_pcm(200, seed=3), 44100, 128, None)["mp3"])
        want = ctx.hide_message(mp3, "fits nowhere")
        pipe = mlib.Pipe(ctx, depth=2, max_job_bytes=8192, scan_threads=1)
        try:
            assert pipe.submit([mp3], ["fits nowhere"]) == 0
            assert pipe.submit([mp3], ["fits nowhere"]) == 1
            assert pipe.submit([mp3], ["fits nowhere"]) is None      # both slots taken
            for t in range(2):
                tk, res = pipe.collect()
                assert tk == t and bytes(res[0]["data"]) == bytes(want["data"])
            assert pipe.stats()["slow"] == 2
        finally:
            pipe.close()
        # the context is usable again after the pipe is gone
        assert bytes(ctx.hide_message(mp3, "fits nowhere")["data"]) == bytes(want["data"])
    finally:
        ctx.close()


def test_chain_check_on_the_device(ctx, mlib):
    """mp3s_chain_resolve_dev against a host walk of the same GrInfo records: cursor, inherited addresses of silent units,
    verdict, per-stream chain ends -- two streams in one batch, silence at a stream start and in the middle"""
    from synth_pcm import synth_pcm
    L = mlib.lib()
    n0, n1 = 300, 53                                   # > 256 frames: more than one workgroup of the scan
    n = n0 + n1
    pcm = np.concatenate([synth_pcm(n0, seed=21), synth_pcm(n1, seed=22)])
    pcm[:5 * 1152] = 0
    pcm[100 * 1152:140 * 1152] = 0
    pcm[n0 * 1152:(n0 + 3) * 1152] = 0                 # the second stream starts in silence
    hdr = np.zeros(n, dtype=mlib.FRAME_HDR_DTYPE)
    hdr["nch"] = 2
    hdr["stream_first"][n0:] = n0
    mdct = ctx.encode_transform(pcm, hdr)
    rf, _ = mlib.rate_frames(44100, 128, 2, n)
    rf["stream"][n0:] = 1
    msg = np.array(mlib.message_frame("the device walks the chain"), dtype=np.uint8)
    hide = np.concatenate([msg, msg])
    rf["hide_end"][:n0] = len(msg)
    rf["hide_end"][n0:] = 2 * len(msg)
    units = n * 4
    segs = np.zeros(2, dtype=mlib.CHAIN_SEG_DTYPE)
    segs["first_frame"], segs["n_frames"] = [0, n0], [n0, n1]
    segs["hide_base"], segs["hide_begin"], segs["hide_end"] = [0, len(msg)], [0, len(msg)], [len(msg), 2 * len(msg)]
    segs["chain_in"][1] = np.arange(16).reshape(4, 4) + 7          # as if the second stream were a block with a carry
    cur = np.zeros(units, dtype=np.int32)
    cur[:n0 * 4] = 3 * np.arange(n0 * 4)
    cur[n0 * 4:] = len(msg) + 3 * np.arange(n1 * 4)
    d_mdct, d_rf, d_hide, d_cur, d_segs = (ctx.to_device(a) for a in (mdct, rf, hide, cur, segs))
    d_ix, d_out, d_en = ctx.alloc(n * 2304 * 2), ctx.alloc(units * 72), ctx.alloc(units * 88)
    d_ver, d_so = ctx.alloc(16), ctx.alloc(2 * 80)
    try:
        mlib.check(L.mp3s_rate_loop_dev(ctx.handle, d_mdct, d_rf, n, d_hide, len(hide), d_cur, None, None, 0, d_ix, d_out, d_en))
        before = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
        mlib.check(L.mp3s_chain_resolve_dev(ctx.handle, d_out, d_rf, n, d_segs, 2, d_cur, None, d_ver, d_so))
        after = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
        ver = ctx.download(d_ver, np.int32, (2,))
        so = ctx.download(d_so, mlib.CHAIN_SEG_OUT_DTYPE, (2,))
    finally:
        for p in (d_mdct, d_rf, d_hide, d_cur, d_segs, d_ix, d_out, d_en, d_ver, d_so):
            ctx.free(p)
    # the host walk (what csrc/mp3s_encode_pipeline.cpp's walk() does)
    exp = before.copy()
    n_redo = 0
    for s in range(2):
        f0, nf = int(segs["first_frame"][s]), int(segs["n_frames"][s])
        c, end = int(segs["hide_begin"][s]), int(segs["hide_end"][s])
        chain = segs["chain_in"][s].copy()
        own = [False] * 4
        used_carry = c < end
        for u in range(f0 * 4, (f0 + nf) * 4):
            k = u & 3
            g = exp[u]
            active = bool(g["flags"] & mlib.RF_ACTIVE)
            used_addr = bool(g["flags"] & mlib.RF_USED_ADDR_IN)
            if not own[k] and (not active or used_addr):
                used_carry = True
            if active:
                own[k] = True
            redo = active and int(cur[u]) != c and min(int(cur[u]), c) < end
            if used_addr and tuple(chain[k][:3]) != (0, 0, 0):
                redo = True
            n_redo += int(redo)
            if active:
                c += int(g["n_tables"])
                chain[k] = list(g["address"]) + [g["quantizer_step"]]
            else:
                exp["address"][u] = chain[k][:3]
                exp["quantizer_step"][u] = chain[k][3]
        assert int(so["cursor"][s]) == c, s
        assert np.array_equal(so["chain"][s], chain), s
        assert bool(so["carry_used"][s]) == used_carry, s
    assert np.array_equal(after, exp)
    assert (before["flags"] & mlib.RF_ACTIVE == 0).sum() >= 40 * 4   # the silent stretches are really silent
    assert int(ver[0]) == n_redo and int(ver[1]) == 0
    assert n_redo > 0                                                  # silence shifts the cursor: the guess 3 per unit fails


def test_pipe_decode_jobs(mlib, golden_dir):
    """decode jobs (MP3 -> WAV bytes + stego bits) interleaved with hide jobs: what mp3s_decode_file gives, file by file"""
    from synth_pcm import synth_pcm
    import frame_synth
    ctx = mlib.Context(0)
    try:
        a = bytes(ctx.encode_pcm(synth_pcm(300, seed=5), 44100, 128, None)["mp3"])
        b = bytes(ctx.encode_pcm(synth_pcm(70, rate=48000, seed=6), 48000, 192, None)["mp3"])
        mono = frame_synth.make_stream(12, 80, mode=3)
        mixed = frame_synth.make_stream(13, 50, block_types=(0, 2), allow_mixed=True)
        test_mp3 = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
        g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
        corpus = [g[n + "__mp3"].tobytes() for n in sorted({k.split("__")[0] for k in g.files})]
        jobs = [("d", [a]), ("d", [b, a]), ("h", [a], ["hello"]), ("d", [mono]), ("d", [mixed, test_mp3]), ("d", [a[:-500]]),   # cut: repeated last frame
                ("d", [mono, a]), ("h", [b], None), ("d", corpus), ("d", [b"junk" * 40, a]), ("d", [test_mp3])]

        def want_of(job):
            if job[0] == "h":
                return ctx.hide_messages(job[1], [None] * len(job[1]) if job[2] is None else job[2])
            out = []
            for f in job[1]:
                try:
                    out.append(ctx.decode_file(f))
                except mlib.Mp3sError as e:
                    out.append(e)
            return out
        want = [want_of(j) for j in jobs]
        for depth, threads in ((3, 2), (1, 1), (4, 4)):
            pipe = mlib.Pipe(ctx, depth=depth, max_job_bytes=1 << 20, scan_threads=threads)
            try:
                got, nxt = [], 0
                seq = jobs * 2
                while len(got) < len(seq):
                    while nxt < len(seq):
                        j = seq[nxt]
                        t = pipe.submit_decode(j[1]) if j[0] == "d" else pipe.submit(j[1], j[2])
                        if t is None:
                            break
                        nxt += 1
                    got.append(pipe.collect()[1])
                st = pipe.stats()
            finally:
                pipe.close()
            for k, res in enumerate(got):
                w = want[k % len(jobs)]
                assert len(res) == len(w), k
                for x, y in zip(res, w):
                    if isinstance(y, Exception):
                        assert isinstance(x, mlib.Mp3sError) and x.code == y.code, (k, x, y)
                        continue
                    assert not isinstance(x, Exception), (k, x)
                    assert bytes(x["data"]) == bytes(y["data"]), k
                    assert (x["kbps"], x["sampling_rate"], x["channels"], x["n_frames"]) == (y["kbps"], y["sampling_rate"], y["channels"], y["n_frames"]), k
                    if jobs[k % len(jobs)][0] == "d":
                        assert np.array_equal(x["bits"], y["bits"]), k
            assert st["fast"] >= 2 * 7, st
    finally:
        ctx.close()


@pytest.mark.gpu
def test_block_jobs_beside_whole_files(ctx, mlib):
    """mp3s_pipe_submit_block: a rank's share of a stream as a job of the pipe, between jobs of whole files; the result is
    mp3s_reencode_block's (bytes, carry, carry_used), the halves put together are the one-call file"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(400, seed=91)
    pcm[190 * 1152:215 * 1152] = 0                           # silence across the middle: the second half looks at its carry
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    other = bytes(ctx.encode_pcm(synth_pcm(150, seed=92), 44100, 128, None)["mp3"])
    for msg in ("hello block jobs", "z" * 300, None):
        whole = ctx.clear_file(mp3) if msg is None else ctx.hide_message(mp3, msg)
        want_other = ctx.hide_message(other, "x")
        for world in (2, 3):
            want = []
            carry = None
            for r in range(world):
                want.append(ctx.reencode_block(mp3, msg, r, world, carry))
                carry = want[-1]["carry_out"]
            pipe = mlib.Pipe(ctx, depth=3, max_job_bytes=1 << 20, scan_threads=2)
            try:
                got, carry = [], None
                for r in range(world):
                    assert pipe.submit([other], ["x"]) is not None
                    assert pipe.submit_block(mp3, msg, r, world, carry) is not None
                    _t, res = pipe.collect()
                    assert bytes(res[0]["data"]) == bytes(want_other["data"])
                    _t, blk = pipe.collect()
                    assert isinstance(blk, dict) and blk["first_frame"] == want[r]["first_frame"] and blk["n_frames"] == want[r]["n_frames"]
                    assert bytes(blk["mp3"]) == want[r]["mp3"], (msg and len(msg), world, r)
                    assert np.array_equal(blk["carry_out"], want[r]["carry_out"]) and blk["is_last"] == want[r]["is_last"]
                    assert blk["hide_offset"] == want[r]["hide_offset"] and blk["too_long"] == want[r]["too_long"]
                    got.append(bytes(blk["mp3"]))
                    carry = blk["carry_out"]
                st = pipe.stats()
            finally:
                pipe.close()
            assert b"".join(got) == bytes(whole["data"])
            assert st["slow"] == 0, st


def test_shallow_pipes_keep_a_job_whose_copy_down_is_still_to_come(mlib):
    """pipes of depth 1 and 2 queue a job's copy down behind the NEXT job's issue (round 4).  If that next job takes the synchronous
    path -- which decodes into the context's PCM buffer -- the copy has to be queued first: a decode job followed by a hide job the
    stages do not take (files of two sampling rates: two device batches) and by ordinary ones, results = the one-call results, for every
    depth (tools/soak_pipe.py found this: 51 wrong decode jobs in 55 000)"""
    import time
    from synth_pcm import synth_pcm
    ctx = mlib.Context(0)
    try:
        a44 = [bytes(ctx.encode_pcm(synth_pcm(n, seed=90 + n), 44100, 128, None)["mp3"]) for n in (300, 500, 120)]
        b48 = bytes(ctx.encode_pcm(synth_pcm(400, rate=48000, seed=95), 48000, 192, None)["mp3"])
        want = {id(f): bytes(ctx.decode_file(f)["data"]) for f in a44 + [b48]}
        hid = {id(f): bytes(ctx.hide_message(f, "after")["data"]) for f in a44 + [b48]}
        for depth in (1, 2, 3):
            for rep in range(3):
                pipe = mlib.Pipe(ctx, depth=depth, max_job_bytes=1 << 21, scan_threads=2)
                try:
                    jobs = [("d", [a44[0]]), ("h", [a44[1], b48]), ("d", [a44[1]]), ("h", [b48, a44[2]]), ("d", [a44[2]]), ("h", [a44[1]]), ("d", [a44[0], a44[1]])]
                    got, nxt = [], 0
                    while len(got) < len(jobs):
                        while nxt < len(jobs):
                            k, files = jobs[nxt]
                            t = pipe.submit_decode(files) if k == "d" else pipe.submit(files, ["after"] * len(files))
                            if t is None:
                                break
                            nxt += 1
                        time.sleep(0.03)                  # (the workers issue what was submitted before the oldest job is asked for)
                        got.append(pipe.collect()[1])
                    for (k, files), res in zip(jobs, got):
                        for f, r in zip(files, res):
                            assert not isinstance(r, Exception)
                            assert bytes(r["data"]) == (want if k == "d" else hid)[id(f)], (depth, rep, k, len(f))
                    assert pipe.stats()["slow"] >= 2
                finally:
                    pipe.close()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_events_as_dispatch_signals_change_no_byte(mlib):
    """MP3S_OPT_PIPE_SIGNALS (the last decode dispatch / the rate loop carry the pipe's events as their own completion signals instead of event
    records behind them) and MP3S_OPT_RATE_SIGNALS + mp3s_ctx_wait_last (the same for a caller that orders contexts itself): stream order only --
    one-file calls, pipe jobs and a hand-ordered device step give the bytes of the default."""
    from synth_pcm import synth_pcm
    base = mlib.Context(0)
    try:
        files = [bytes(base.encode_pcm(synth_pcm(n, seed=300 + n), 44100, 128, None)["mp3"]) for n in (2500, 700, 90)]
        want_h = [bytes(base.hide_message(f, "signals")["data"]) for f in files]
        want_d = [bytes(base.decode_file(f)["data"]) for f in files]
    finally:
        base.close()
    for v in (1, 2, 3):
        c = mlib.Context(0)
        try:
            c.set_option("pipe_signals", v)
            for rep in range(2):
                for f, h, d in zip(files, want_h, want_d):
                    assert bytes(c.hide_message(f, "signals")["data"]) == h, (v, len(f))
                    assert bytes(c.decode_file(f)["data"]) == d, (v, len(f))
            pipe = mlib.Pipe(c, depth=3, max_job_bytes=1 << 22, scan_threads=2)
            try:
                assert pipe.submit(files, ["signals"] * len(files)) is not None
                for rep in range(3):
                    assert pipe.submit_decode(files) is not None
                    got_h = pipe.collect()[1]
                    assert pipe.submit(files, ["signals"] * len(files)) is not None
                    got_d = pipe.collect()[1]
                    assert [bytes(r["data"]) for r in got_h] == want_h and [bytes(r["data"]) for r in got_d] == want_d, v
                pipe.collect()
            finally:
                pipe.close()
        finally:
            c.close()
    # the rate loop's own signal for a caller's contexts: a second context's copy waits for it with wait_last
    a, b = mlib.Context(0), mlib.Context(0)
    try:
        a.set_option("rate_signals", 1)
        pcm = synth_pcm(400, seed=77)
        r0 = a.encode_pcm(pcm, 44100, 128, None)
        b.wait_last(a)                       # (whatever a's last rate loop was: nothing of b starts before it)
        r1 = b.encode_pcm(pcm, 44100, 128, None)
        assert bytes(r0["mp3"]) == bytes(r1["mp3"])
    finally:
        a.close(); b.close()
