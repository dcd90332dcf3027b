"""ONE file as chunks through the overlapped stages: mp3s_hide_message / mp3s_clear_file /
mp3s_decode_file / mp3s_decode_stream walk the frame headers a chunk at a time and keep several chunks in flight; the
result must be the bytes of the same call with the stages one after the other (MP3S_OPT_FILE_PIPELINE = 0: round 2's
path, which tests/test_gpu_dropin.py and tests/test_gpu_parity.py pin to the reference's goldens and to the oracle).
Reference call shape: steganography.py:137-162, decoder/MP3_Parser.py:68-80, encoder/MP3_Encoder.py:607-609."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class options:
    def __init__(self, ctx, **kw):
        self.ctx, self.kw = ctx, kw

    def __enter__(self):
        self.old = {k: self.ctx.set_option(k, v) for k, v in self.kw.items()}

    def __exit__(self, *a):
        for k, v in self.old.items():
            self.ctx.set_option(k, v)


def legacy(ctx, f, *a):
    with options(ctx, file_pipeline=0):
        return f(*a)


def same_file(a, b):
    return bytes(a["data"]) == bytes(b["data"]) and all(a[k] == b[k] for k in ("too_long", "hide_offset", "n_frames", "kbps", "sampling_rate"))


def test_goldens_through_chunks(ctx, mlib, golden_dir):
    """tests/test.mp3 (36 frames): hide / clear / decode in chunks of 4 ... 36 frames = the reference's bytes"""
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    want_f64 = bytes(np.load(os.path.join(golden_dir, "g2_decode_testmp3.npz"))["pcm_sha256"]).decode()     # the reference's float64 PCM
    want_hide, want_clear, want_wav = legacy(ctx, ctx.hide_message, data, "ddd"), legacy(ctx, ctx.clear_file, data), legacy(ctx, ctx.decode_file, data)
    import json
    facade = json.load(open(os.path.join(golden_dir, "g3_facade.json")))                                       # the reference's own outputs
    assert hashlib.sha256(bytes(want_hide["data"])).hexdigest() == facade["hide_sha256"]
    assert hashlib.sha256(bytes(want_wav["data"])).hexdigest() == "d6787aaab67826a07221eb6cb9d604c44bf3e7a1d79580ec7edded5dc17c6a90"
    s0 = ctx.run_stats()
    for chunk in (4, 5, 7, 16, 35, 36, 0):
        with options(ctx, chunk_frames=chunk):
            assert same_file(ctx.hide_message(data, "ddd"), want_hide), chunk
            assert same_file(ctx.clear_file(data), want_clear), chunk
            w = ctx.decode_file(data)
            assert bytes(w["data"]) == bytes(want_wav["data"]) and np.array_equal(w["bits"], want_wav["bits"]), chunk
            f64 = ctx.decode_stream(data, mlib.MP3S_PCM_F64)
            assert hashlib.sha256(f64["pcm"].tobytes()).hexdigest() == want_f64, chunk
    s1 = ctx.run_stats()
    assert s1["files"] - s0["files"] == 7 * 4 and s1["fallbacks"] == s0["fallbacks"], (s0, s1)
    assert s1["chunks"] - s0["chunks"] > 7 * 4 * 2


def test_float_formats_in_chunks_are_bit_identical(ctx, mlib, golden_dir):
    import frame_synth
    streams = [open(os.path.join(golden_dir, "test.mp3"), "rb").read(),
               frame_synth.make_stream(7, 300, block_types=(0, 1, 2, 3), use_reservoir=True),           # reservoir, all block types
               frame_synth.make_stream(8, 200, mode=3, use_reservoir=True),                              # mono
               frame_synth.make_stream(12, 150, mode=1, block_types=(0, 2))]                            # joint stereo, short blocks
    for si, mp3 in enumerate(streams):
        for fmt in (mlib.MP3S_PCM_F64, mlib.MP3S_PCM_F32, mlib.MP3S_PCM_I16):
            want = legacy(ctx, ctx.decode_stream, mp3, fmt)
            for chunk in (9, 64, 0):
                with options(ctx, chunk_frames=chunk):
                    got = ctx.decode_stream(mp3, fmt)
                assert got["pcm"].tobytes() == want["pcm"].tobytes(), (si, fmt, chunk)
                assert np.array_equal(got["bits"], want["bits"]) and got["n_frames"] == want["n_frames"], (si, fmt, chunk)
                assert (got["channels"], got["sampling_rate"], got["bit_rate"]) == (want["channels"], want["sampling_rate"], want["bit_rate"])


def test_carries_across_chunks(ctx, mlib, orc):
    """silences at chunk boundaries (inherited addresses: SURVEY E7), messages that end inside a later chunk (the cursor is
    live at a boundary), a truncated last frame -- chunks that depend on their carry are run again, the bytes stay"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(700, seed=51)
    pcm[:40 * 1152] = 0                                      # the stream starts in silence
    pcm[120 * 1152:136 * 1152] = 0                           # silences that straddle chunk boundaries of 64 and 128 frames
    pcm[250 * 1152:262 * 1152] = 0
    pcm[500 * 1152:530 * 1152, 1] = 0                        # one channel silent
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    s0 = ctx.run_stats()
    for data in (mp3, mp3[:-3]):
        for msg in ("short", "x" * 150, "y" * 400, None):
            want = legacy(ctx, ctx.clear_file, data) if msg is None else legacy(ctx, ctx.hide_message, data, msg)
            for chunk in (64, 128, 250, 0):
                with options(ctx, chunk_frames=chunk):
                    got = ctx.clear_file(data) if msg is None else ctx.hide_message(data, msg)
                assert same_file(got, want), (len(data), msg and len(msg), chunk)
    s1 = ctx.run_stats()
    assert s1["files"] > s0["files"] and s1["reruns"] > s0["reruns"], (s0, s1)      # some chunk did depend on its carry
    # against the oracle directly
    d = orc.decode(mp3)
    o = orc.encode(orc.pcm_to_i16(d["pcm"]), 44100, 128, np.array(mlib.message_frame("y" * 400)))
    with options(ctx, chunk_frames=64):
        assert bytes(ctx.hide_message(mp3, "y" * 400)["data"]) == o["mp3"]


def test_what_the_chunks_do_not_take_falls_back(ctx, mlib, golden_dir):
    import frame_synth
    mixed = frame_synth.make_stream(9, 60, block_types=(0, 2), allow_mixed=True)      # scalefactors inherited across frames
    mono = frame_synth.make_stream(8, 50, mode=3)
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    tail = data + b"\x00" * 700                                                       # a bad header: the last frame is repeated (D12)
    with options(ctx, chunk_frames=16):
        # scalefactors inherited across frames: the file goes through the chunks all the same (round 4: its side records and main
        # data lie in file-wide arrays, the Huffman kernel's walk back crosses the chunk boundaries) ...
        s0 = ctx.run_stats()
        assert legacy(ctx, ctx.decode_stream, mixed, mlib.MP3S_PCM_F64)["pcm"].tobytes() == ctx.decode_stream(mixed, mlib.MP3S_PCM_F64)["pcm"].tobytes()
        s1 = ctx.run_stats()
        assert s1["files"] - s0["files"] == 1 and s1["chunks"] - s0["chunks"] == 4 and s1["fallbacks"] == s0["fallbacks"], (s0, s1)
        # ... and without those arrays (each chunk's piece of the file uploaded by the caller, as for files above 1 GB) once more
        # in one piece -- not through the host parser
        with options(ctx, file_up=0):
            s0 = ctx.run_stats()
            assert legacy(ctx, ctx.decode_stream, mixed, mlib.MP3S_PCM_F64)["pcm"].tobytes() == ctx.decode_stream(mixed, mlib.MP3S_PCM_F64)["pcm"].tobytes()
            s1 = ctx.run_stats()
            assert s1["files"] - s0["files"] == 1 and s1["chunks"] - s0["chunks"] == 1 and s1["fallbacks"] == s0["fallbacks"], (s0, s1)
        joint = frame_synth.make_stream(10, 80, mode=1, mode_ext=2, block_types=(0, 2), allow_mixed=True)      # the same for a re-encode
        assert same_file(legacy(ctx, ctx.hide_message, joint, "mixed blocks"), ctx.hide_message(joint, "mixed blocks"))
        s0 = ctx.run_stats()
        assert bytes(legacy(ctx, ctx.decode_file, tail)["data"]) == bytes(ctx.decode_file(tail)["data"])
        assert same_file(legacy(ctx, ctx.hide_message, tail, "abc"), ctx.hide_message(tail, "abc"))
        with pytest.raises(mlib.Mp3sError) as e1:
            ctx.hide_message(mono, "abc")
        assert e1.value.code == mlib.E_UNSUPPORTED
        assert ctx.decode_file(b"\x00" * 10)["n_frames"] == 0                         # no sync: an empty WAV, as the reference writes
        s1 = ctx.run_stats()
        assert s1["fallbacks"] - s0["fallbacks"] >= 3, (s0, s1)


def test_long_file_automatic_chunks_and_the_facade(ctx, mlib, tmp_path):
    """a 12 000-frame file with the automatic plan (a short first chunk, then the rest), through Steganography.hide_message"""
    from synth_pcm import synth_pcm
    from mp3stego import Steganography
    pcm = synth_pcm(12000, seed=61)
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    want = legacy(ctx, ctx.hide_message, mp3, "The quick brown fox")
    s0 = ctx.run_stats()
    got = ctx.hide_message(mp3, "The quick brown fox")
    s1 = ctx.run_stats()
    assert same_file(got, want)
    assert s1["files"] - s0["files"] == 1 and s1["chunks"] - s0["chunks"] >= 2 and s1["fallbacks"] == s0["fallbacks"], (s0, s1)
    wav_want = legacy(ctx, ctx.decode_file, mp3)
    wav = ctx.decode_file(mp3)
    assert bytes(wav["data"]) == bytes(wav_want["data"]) and np.array_equal(wav["bits"], wav_want["bits"])
    src, dst = tmp_path / "in.mp3", tmp_path / "out.mp3"
    src.write_bytes(mp3)
    st = Steganography(quiet=True)
    assert st.hide_message(str(src), str(dst), "The quick brown fox") is False
    assert dst.read_bytes() == bytes(want["data"])
    got = ctx.hide_message_chunked(mp3, "The quick brown fox", 3000)
    assert same_file(got, want)


def test_file_from_the_helper_thread_or_chunk_by_chunk(ctx, mlib, golden_dir):
    """the whole file uploaded in pieces by the context's helper thread (the default) or every chunk's piece by the calling
    thread (MP3S_OPT_FILE_UP = 0: the path of files above 1 GB): same bytes, on files of one chunk and of several, cut files, and
    a decode; steganography.py:137-162"""
    from synth_pcm import synth_pcm
    mp3 = bytes(ctx.encode_pcm(synth_pcm(5000, seed=71), 44100, 128, None)["mp3"])
    small = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    for data in (mp3, mp3[:-777], small, small[:-5]):
        for chunk in (0, 700):
            with options(ctx, chunk_frames=chunk):
                with options(ctx, file_up=1):
                    a, ca, wa = ctx.hide_message(data, "helper thread"), ctx.clear_file(data), ctx.decode_file(data)
                with options(ctx, file_up=0):
                    b, cb, wb = ctx.hide_message(data, "helper thread"), ctx.clear_file(data), ctx.decode_file(data)
            assert same_file(a, b) and same_file(ca, cb) and bytes(wa["data"]) == bytes(wb["data"]), (len(data), chunk)
            assert same_file(a, legacy(ctx, ctx.hide_message, data, "helper thread"))


@pytest.mark.gpu
def test_a_failing_chunk_leaves_the_context_usable(ctx, mlib):
    """a hard error in the middle of a one-file call (MP3S_OPT_FAIL_CHUNK: chunk k fails with MP3S_E_HIP while its front end and
    the chunks in front of it are in flight) must leave nothing queued and no slot taken: the error comes back, and the next
    calls on the same context -- hide, clear, decode, in chunks -- give the bytes they always give (run_file's one cleanup
    path; steganography.py:137-162)"""
    from synth_pcm import synth_pcm
    mp3 = bytes(ctx.encode_pcm(synth_pcm(3000, seed=72), 44100, 128, None)["mp3"])
    ref = legacy(ctx, ctx.hide_message, mp3, "after the failure")
    ref_clear = legacy(ctx, ctx.clear_file, mp3)
    ref_wav = bytes(legacy(ctx, ctx.decode_file, mp3)["data"])
    with options(ctx, chunk_frames=500):
        for k in (1, 2, 4, 6):
            for call in (lambda: ctx.hide_message(mp3, "after the failure"), lambda: ctx.decode_file(mp3)):
                s0 = ctx.run_stats()
                ctx.set_option("fail_chunk", k)
                with pytest.raises(mlib.Mp3sError) as e:
                    call()
                assert e.value.code == mlib.E_HIP and "MP3S_OPT_FAIL_CHUNK" in str(e.value)
                assert ctx.get_option("fail_chunk") == 0                  # it fires once
                s1 = ctx.run_stats()
                assert s1["files"] == s0["files"] and s1["fallbacks"] == s0["fallbacks"]
                assert same_file(ctx.hide_message(mp3, "after the failure"), ref)
                assert same_file(ctx.clear_file(mp3), ref_clear)
                assert bytes(ctx.decode_file(mp3)["data"]) == ref_wav
                assert ctx.run_stats()["files"] == s1["files"] + 3


@pytest.mark.gpu
def test_inherited_scalefactors_across_chunks(ctx, mlib, orc, golden_dir):
    """streams whose granules read scalefactors written many frames earlier (mixed blocks, scfsi behind a short granule 0:
    decoder/FrameSideInformation.py:11-37, decoder/Frame.py:392-405, 423-437) as chunks of 9 and 64 frames: the granule that wrote an
    entry last may lie several chunks back.  float64 PCM = the oracle's (and the reference's, for the g7 streams), stego bits
    and re-encodes = the one-piece call's; every file really went through several chunks"""
    import frame_synth
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    streams = [(n, g[n + "__mp3"].tobytes()) for n in names if g[n + "__si_mixed_block_flag"].any()]
    assert streams
    # long ones: short / long switches and mixed blocks, mono / joint stereo, the inherited entries up to dozens of frames back
    for i in range(4):
        streams.append(("synth%d" % i, frame_synth.make_stream(50 + i, 200 + 37 * i, block_types=(0, 1, 2, 3), allow_mixed=True, mode=(0, 1, 3, 0)[i],
                                                              mode_ext=2 if i == 1 else 0, use_reservoir=i >= 2)))
    for name, data in streams:
        want = orc.decode(data)
        whole_bits = legacy(ctx, ctx.decode_stream, data, mlib.MP3S_PCM_F64)["bits"]
        n = want["n_frames"]
        for chunk in (9, 64):
            if n <= chunk:
                continue
            with options(ctx, chunk_frames=chunk):
                s0 = ctx.run_stats()
                got = ctx.decode_stream(data, mlib.MP3S_PCM_F64)
                s1 = ctx.run_stats()
                assert got["pcm"].tobytes() == want["pcm"].tobytes(), (name, chunk)
                assert np.array_equal(got["bits"], whole_bits), (name, chunk)
                assert s1["files"] - s0["files"] == 1 and s1["chunks"] - s0["chunks"] == -(-n // chunk) and s1["fallbacks"] == s0["fallbacks"], (name, chunk, s0, s1)
                if want["channels"] == 2 and want["sampling_rate"] in (32000, 44100, 48000):
                    try:
                        ref = legacy(ctx, ctx.hide_message, data, "across the chunks")
                    except mlib.Mp3sError:
                        continue                                # (a bit rate the encoder does not take: refused by both paths)
                    assert same_file(ctx.hide_message(data, "across the chunks"), ref), (name, chunk)


@pytest.mark.gpu
def test_every_context_of_a_process_gets_its_overlap(mlib):
    """three contexts in one process, each with its own pipe (made by its first one-file call): the streams of a pipe are chosen by
    a rehearsal judged against the same miniature on ONE stream -- at most 12 miniatures and 15 ms, reported in
    mp3s_ctx_run_stats -- and no context may end up on lanes that share hardware queues: the same 2 500-frame call takes the same
    time on all three (within 30 %; round 3's first context of three ran at half speed before the rehearsal existed)"""
    import time
    from synth_pcm import synth_pcm
    ctxs = [mlib.Context(0) for _ in range(3)]
    try:
        mp3 = bytes(ctxs[0].encode_pcm(synth_pcm(2500, seed=73), 44100, 128, None)["mp3"])
        ref = None
        med = []
        for c in ctxs:
            out = bytes(c.hide_message(mp3, "three contexts")["data"])          # the first call makes the pipe
            ref = ref or out
            assert out == ref
            st = c.run_stats()
            assert 3 <= st["rehearsals"] <= 12 and 0 < st["rehearsal_us"] < 70000, st    # (15 ms of budget, or 13 one-stream miniatures within 60 ms where the process is slow, + the miniature that started inside it + allocation)
            assert st["queue_shared"] == 0, st
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                r = c.hide_message(mp3, "three contexts"); del r
                ts.append(time.perf_counter() - t0)
            med.append(sorted(ts)[len(ts) // 2])
        # (what the rehearsal guards against is a context at HALF speed -- 1.08 - 1.18 ms against 0.70 in round 3; from run to run the
        # contexts of a process differ by up to 20 %: the first stream a process makes is not like the others -- and by 35 % on two boxes
        # of round 4's last day: 0.61 / 0.58 / 0.79 ms; the bound stays below the 1.54 the half-speed case had at its best)
        assert max(med) <= 1.45 * min(med), [round(m * 1e3, 3) for m in med]
        # a user's pipe on the first context: its own rehearsal (the context's stream is the same: decided earlier, nothing rehearsed)
        p = mlib.Pipe(ctxs[0], depth=2, max_job_bytes=len(mp3) + 65536, scan_threads=1)
        ps = p.stats()
        assert ps["rehearsals"] <= 12 and ps["queue_shared"] == 0, ps
        p.close()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_result_written_to_a_file_descriptor_chunk_by_chunk(ctx, mlib, orc, golden_dir, tmp_path):
    """mp3s_hide_message_fd / mp3s_clear_file_fd (what Steganography.hide_message / clear_file end in, reference steganography.py:137-182,
    encoder/encoder.py:53-57): the file holds exactly the bytes the call returns otherwise -- over a longer file that was there, with chunks
    that depend on their carry (run again: written at the end), through the paths the chunks do not take, in place through the facade, and
    a stream the call refuses leaves the output as it was."""
    import frame_synth
    from synth_pcm import synth_pcm
    from mp3stego import Steganography
    pcm = synth_pcm(700, seed=51)
    pcm[:40 * 1152] = 0
    pcm[120 * 1152:136 * 1152] = 0                           # silences that straddle chunk boundaries (inherited addresses: chunks run again)
    pcm[250 * 1152:262 * 1152] = 0
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    out = tmp_path / "out.mp3"

    def through_fd(data, msg, fresh=False):
        # something longer is there: it is overwritten and cut to length at the end of the call -- or nothing: then chunk by chunk
        out.write_bytes(b"" if fresh else b"\xaa" * (len(data) + 5000))
        fd = os.open(str(out), os.O_WRONLY)
        try:
            r = ctx.recode_to_fd(data, msg, fd)
            assert os.lseek(fd, 0, os.SEEK_CUR) == 0         # pwrite: the descriptor's position is not moved
        finally:
            os.close(fd)
        got = out.read_bytes()
        assert len(got) == r["len"]
        return got, r

    s0 = ctx.run_stats()
    for data in (mp3, mp3[:-3]):
        for msg in ("short", "y" * 400, None):
            want = ctx.clear_file(data) if msg is None else ctx.hide_message(data, msg)
            for chunk in (64, 250, 0):
                for fresh in (False, True):
                    with options(ctx, chunk_frames=chunk):
                        got, r = through_fd(data, msg, fresh)
                    assert got == bytes(want["data"]), (len(data), msg and len(msg), chunk, fresh)
                assert r["too_long"] == want["too_long"] and r["hide_offset"] == want["hide_offset"] and r["n_frames"] == want["n_frames"]
    s1 = ctx.run_stats()
    assert s1["reruns"] > s0["reruns"]                                               # some chunk was run again on its real carry
    assert through_fd(mp3, "y" * 400)[0] == orc.encode(orc.pcm_to_i16(orc.decode(mp3)["pcm"]), 44100, 128, np.array(mlib.message_frame("y" * 400)))["mp3"]
    # what the chunks do not take (a repeated last frame; mixed blocks without the file-wide arrays)
    tail = open(os.path.join(golden_dir, "test.mp3"), "rb").read() + b"\x00" * 700
    assert through_fd(tail, "abc", True)[0] == bytes(ctx.hide_message(tail, "abc")["data"])
    joint = frame_synth.make_stream(10, 80, mode=1, mode_ext=2, block_types=(0, 2), allow_mixed=True)
    with options(ctx, chunk_frames=16, file_up=0):
        assert through_fd(joint, None, True)[0] == bytes(ctx.clear_file(joint)["data"])
    # a frame damaged in the last chunk: found on the device when that chunk comes down -- a file that held something is as it was
    bad = bytearray(mp3)
    refs = mlib.walk_stream(mp3)["refs"]
    at = int(refs["file_off"][-3]) + 40
    bad[at:at + 60] = b"\xff" * 60
    bad = bytes(bad)
    try:
        want_bad = bytes(ctx.hide_message(bad, "abc")["data"])
    except mlib.Mp3sError:
        want_bad = None
    out.write_bytes(b"kept" * 1000)
    fd = os.open(str(out), os.O_WRONLY)
    try:
        with options(ctx, chunk_frames=64):
            try:
                r = ctx.recode_to_fd(bad, "abc", fd)
                assert want_bad is not None and out.read_bytes() == want_bad and r["len"] == len(want_bad)
            except mlib.Mp3sError:
                assert want_bad is None and out.read_bytes() == b"kept" * 1000
    finally:
        os.close(fd)
    # decode: the WAV into a file, fresh (chunk by chunk) or over something longer
    wav_want = ctx.decode_file(mp3)
    for fresh in (False, True):
        out.write_bytes(b"" if fresh else b"\xbb" * (len(wav_want["data"]) + 999))
        fd = os.open(str(out), os.O_WRONLY)
        try:
            with options(ctx, chunk_frames=64):
                r = ctx.decode_file_to_fd(mp3, fd)
        finally:
            os.close(fd)
        assert out.read_bytes() == bytes(wav_want["data"]) and np.array_equal(r["bits"], wav_want["bits"]) and r["kbps"] == wav_want["kbps"]
    assert ctx.decode_file_to_fd(b"\x00" * 10, os.open(str(out), os.O_WRONLY))["n_frames"] == 0 and len(out.read_bytes()) == 44
    # a stream the call refuses: the error of the other entry point, and through the facade the output file is as it was
    mono = frame_synth.make_stream(8, 50, mode=3)
    fd = os.open(str(out), os.O_WRONLY)
    try:
        with pytest.raises(mlib.Mp3sError) as e1:
            ctx.recode_to_fd(mono, "abc", fd)
        assert e1.value.code == mlib.E_UNSUPPORTED
    finally:
        os.close(fd)
    # the facade: in place (the result replaces the input only when the input has been read), and into a new file
    st = Steganography(quiet=True)
    src = tmp_path / "in.mp3"
    src.write_bytes(mp3)
    want = bytes(ctx.hide_message(mp3, "in place")["data"])
    with options(ctx, chunk_frames=64):
        assert st.hide_message(str(src), str(src), "in place") is False
    assert src.read_bytes() == want
    src.write_bytes(mp3)
    new = tmp_path / "new.mp3"
    st.clear_file(str(src), str(new))
    assert new.read_bytes() == bytes(ctx.clear_file(mp3)["data"])


@pytest.mark.gpu
def test_files_with_larger_frames_park_the_pipe_and_change_nothing(mlib):
    """A context's own pipe is sized by the largest frame it has seen; a file with larger frames (a higher bit rate) makes a new one and the old one
    is PARKED until the context goes (csrc/run_file.cpp ensure_own_pipe: freeing its buffers at that point halved the rate of the next ten calls'
    copies -- DESIGN 6).  Files of growing and shrinking frame sizes through one context, hide and decode: every result equals the stages one
    after the other, none of the calls falls back, and the context can be destroyed with three pipes parked."""
    from synth_pcm import synth_pcm
    ctx = mlib.Context(0)
    try:
        pcm = {r: synth_pcm(2600, seed=0x51 + r, rate=r) for r in (32000, 44100, 48000)}
        order = [(44100, 64), (44100, 128), (48000, 192), (44100, 128), (44100, 320), (32000, 64), (48000, 320), (44100, 64)]   # 208 .. 1 044 bytes per frame, up and down
        files = {rk: bytes(ctx.encode_pcm(pcm[rk[0]], rk[0], rk[1], None)["mp3"]) for rk in set(order)}
        rs0 = ctx.run_stats()
        for i, rk in enumerate(order):
            data = files[rk]
            got = ctx.hide_message(data, "frames of %d bytes" % (len(data) // 2600))
            want = legacy(ctx, ctx.hide_message, data, "frames of %d bytes" % (len(data) // 2600))
            assert same_file(got, want), (i, rk)
            d16, w16 = ctx.decode_stream(data, mlib.MP3S_PCM_I16), legacy(ctx, ctx.decode_stream, data, mlib.MP3S_PCM_I16)
            assert np.array_equal(d16["pcm"], w16["pcm"]) and np.array_equal(d16["bits"], w16["bits"]), (i, rk)
            del got, want, d16, w16
        rs1 = ctx.run_stats()
        assert rs1["fallbacks"] == rs0["fallbacks"] and rs1["files"] - rs0["files"] == 2 * len(order)
    finally:
        ctx.close()
