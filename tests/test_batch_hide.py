"""SURVEY 8f n4, encode side: hide / clear over a list of files as one device batch per (sampling rate, bitrate).
Every stream of the batch against the oracle (decode -> int16 -> encode with the framed message), against the
single-file entry points, and the per-file status of files that cannot be re-encoded."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expect(orc, mlib, mp3, message):
    d = orc.decode(mp3)
    pcm = orc.pcm_to_i16(d["pcm"])
    bits = None if message is None else np.array(mlib.message_frame(message))
    return orc.encode(pcm, int(d["sampling_rate"]), int(d["bit_rate"]) // 1000, bits)


def test_hide_messages_batch_matches_oracle_and_single_calls(ctx, mlib, orc, golden_dir):
    from synth_pcm import synth_pcm
    rng = np.random.default_rng(11)
    files, msgs = [], []
    # three (rate, bitrate) groups, streams of 1 .. 260 frames, with silence in the middle of one (stale addresses, E7)
    for i, (rate, kbps, n) in enumerate([(44100, 128, 60), (48000, 192, 35), (44100, 128, 1), (32000, 64, 90),
                                         (44100, 128, 260), (48000, 192, 2), (44100, 128, 17), (32000, 64, 5)]):
        pcm = synth_pcm(n, rate=rate, seed=1000 + i)
        if n > 100:
            pcm[50 * 1152:70 * 1152] = 0
        files.append(ctx.encode_pcm(pcm, rate, kbps, None)["mp3"])
    g6 = np.load(os.path.join(golden_dir, "g6_synth128.npz"))["mp3"].tobytes()
    files.append(g6)
    msgs = ["short", None, "x", "a message that does not fit into five frames " * 40, "", "ab", None, "héllo wörld ✓",
            "".join(chr(int(c)) for c in rng.integers(32, 127, size=400))]     # > 1024 bits: the variant path inside a batch
    out = ctx.hide_messages(files, msgs)
    assert len(out) == len(files)
    for i, (f, m, r) in enumerate(zip(files, msgs, out)):
        assert not isinstance(r, Exception), (i, r)
        o = _expect(orc, mlib, f, m)
        assert o["rc"] == 0 and r["data"] == o["mp3"], i
        assert r["hide_offset"] == o["hide_offset"] and r["too_long"] == bool(o["too_long"]), i
        single = ctx.clear_file(f) if m is None else ctx.hide_message(f, m)
        assert single["data"] == r["data"] and single["too_long"] == r["too_long"], i
    # what went in comes out (the reveal slices by characters of the latin-1 reading: compare on ASCII messages)
    assert mlib.reveal_message(out[0]["data"])["data"] == b"short"
    got = bytes(mlib.reveal_message(out[8]["data"])["data"])
    k = len(msgs[8]) if not out[8]["too_long"] else int(out[8]["hide_offset"]) // 8 - len("400#")
    assert k > 50 and got[:k] == msgs[8].encode()[:k]
    assert out[3]["too_long"] and not out[0]["too_long"]


def test_hide_messages_reports_each_file(ctx, mlib, orc, golden_dir):
    """files the reference could not re-encode get their own status (the code the single-file call fails with); the
    rest of the batch is unaffected"""
    from synth_pcm import synth_pcm
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    good = ctx.encode_pcm(synth_pcm(12, seed=5), 44100, 128, None)["mp3"]
    files = [good, b"", b"not an mp3 file at all" * 10, good[:-1000], good]
    msgs = ["one", "two", "three", "four", None]
    for n in names:                                                  # mono, MS stereo, reservoir, mixed blocks ...
        files.append(g[n + "__mp3"].tobytes())
        msgs.append("m" + n)
    out = ctx.hide_messages(files, msgs)
    for i, (f, m, r) in enumerate(zip(files, msgs, out)):
        try:
            single = ctx.clear_file(f) if m is None else ctx.hide_message(f, m)
        except mlib.Mp3sError as e:
            assert isinstance(r, mlib.Mp3sError) and r.code == e.code, (i, r, e)
            continue
        assert not isinstance(r, Exception), (i, r)
        assert r["data"] == single["data"] and r["too_long"] == single["too_long"], i
        o = _expect(orc, mlib, f, m)
        assert o["rc"] == 0 and r["data"] == o["mp3"], i
    assert isinstance(out[1], mlib.Mp3sError) and isinstance(out[2], mlib.Mp3sError)
    assert not isinstance(out[0], Exception) and not isinstance(out[3], Exception) and not isinstance(out[4], Exception)
    # status == NULL in the C call: the first failing file fails the call
    import ctypes as C
    L = mlib.lib()
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files[:3]]
    ptr = (C.c_void_p * 3)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * 3)(*[len(b) for b in bufs])
    res, owner = (mlib.File * 3)(), C.c_void_p()
    assert L.mp3s_hide_messages(ctx.handle, ptr, lens, 3, None, None, C.byref(owner), res, None) == out[1].code
    assert not owner.value


def test_many_short_files_one_batch(ctx, mlib, orc):
    """the shape the entry point is for: a few hundred files of a few dozen frames; a sample of them against the oracle,
    all of them against a checksum of the single-file path"""
    import zlib
    from synth_pcm import synth_pcm
    files, msgs = [], []
    for i in range(120):
        n = 8 + (i * 7) % 40
        files.append(ctx.encode_pcm(synth_pcm(n, seed=4000 + i), 44100, 128, None)["mp3"])
        msgs.append(None if i % 10 == 9 else "file %d carries this" % i)
    out = ctx.hide_messages(files, msgs)
    crc_batch = 0
    crc_single = 0
    fitted = 0
    for i, (f, m, r) in enumerate(zip(files, msgs, out)):
        assert not isinstance(r, Exception), (i, r)
        crc_batch = zlib.crc32(r["data"], crc_batch)
        single = ctx.clear_file(f) if m is None else ctx.hide_message(f, m)
        crc_single = zlib.crc32(single["data"], crc_single)
        if i % 17 == 0:
            assert r["data"] == _expect(orc, mlib, f, m)["mp3"], i
        if m is not None:                                            # ~12 bits per frame: the shortest files cut it
            # (and "too long" is hide_offset < n_bits - 1 in the reference: the very last bit may be missing unflagged)
            k = min(len(m), max(0, int(r["hide_offset"]) // 8 - len("%d#" % len(m))))
            assert bytes(mlib.reveal_message(r["data"])["data"])[:k] == m.encode()[:k], i
            fitted += not r["too_long"]
    assert crc_batch == crc_single and fitted > 60


def test_two_threads_two_contexts(mlib, orc):
    """calls on one context are serialised by the caller, different contexts run concurrently (ctypes
    releases the GIL): two threads, each with its own context on the same device, must not disturb each other (shared
    state = the page-locked block cache, the constant tables, the per-thread error text)"""
    import threading
    from synth_pcm import synth_pcm
    streams = []
    boot = mlib.Context(0)
    for i in range(6):
        streams.append(boot.encode_pcm(synth_pcm(20 + 7 * i, seed=600 + i), 44100, 128, None)["mp3"])
    want = [bytes(boot.hide_message(s, "thread %d" % i)["data"]) for i, s in enumerate(streams)]
    boot.close()
    errors = []

    def work(tid):
        try:
            ctx = mlib.Context(0)
            for rnd in range(15):
                for i, s in enumerate(streams):
                    if (i + tid) % 2:
                        continue
                    if bytes(ctx.hide_message(s, "thread %d" % i)["data"]) != want[i]:
                        errors.append((tid, rnd, i))
                    try:
                        ctx.hide_message(b"\x00" * 64, "x")          # an error text per thread
                    except mlib.Mp3sError:
                        pass
                d = ctx.decode_stream(streams[tid], mlib.MP3S_PCM_I16)
                if d["n_frames"] != 20 + 7 * tid:
                    errors.append((tid, rnd, "decode"))
            ctx.close()
        except Exception as e:                                       # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_damaged_member_of_a_batch_is_refused_as_the_single_call_refuses_it(ctx, mlib, orc, golden_dir):
    """A stream whose frames inherit scalefactors (it goes to the host parser as a whole) with main data the parser
    rejects half way: the batch entry points run a failing group's files again one by one, and the record the failed
    parse left behind once let such a file through as a shorter one (found by tools/soak_mutant_hide.py)."""
    import test_fuzz as tf
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    refused = 0
    for seed in (30120, 30150, 34088):
        rng = np.random.default_rng(seed)
        src = g[names[seed % len(names)] + "__mp3"].tobytes()
        gen = tf.header_mutants(mlib, src, 8, seed) if seed % 3 == 0 else tf.mutants(src, 8, seed)
        files = []
        for m in gen:
            if rng.integers(0, 4) == 0:
                m = m[:len(m) - int(rng.integers(1, 900))]
            files.append(m)
        msgs = [None if rng.integers(0, 4) == 0 else "m" * int(rng.integers(0, 60)) for _ in files]
        out = ctx.hide_messages(files, msgs)
        for i, (f, msg, r) in enumerate(zip(files, msgs, out)):
            try:
                single = ctx.clear_file(f) if msg is None else ctx.hide_message(f, msg)
            except mlib.Mp3sError:
                single = None
            if single is None:
                assert isinstance(r, Exception), (seed, i)
                refused += 1
            else:
                assert not isinstance(r, Exception) and r["data"] == single["data"], (seed, i)
    assert refused >= 3
