"""Decode-side corpus the reference encoder cannot produce (BASELINE config 5): short / start / stop /
mixed blocks, MS stereo, mono, CRC, bit reservoir, ID3 tag, code books 4/14, PCM far outside [-1, 1].
Streams come from tests/frame_synth.py; the expected output is the reference's own decode (gen_golden.py corpus)."""
import hashlib
import os

import numpy as np
import pytest


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def corpus(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    return g, names


def test_synthesiser_is_reproducible(corpus):
    """the committed streams are what tests/frame_synth.py generates (seeded)"""
    import frame_synth
    g, names = corpus
    fresh = frame_synth.corpus()
    assert sorted(fresh) == names
    for n in names:
        assert fresh[n] == g[n + "__mp3"].tobytes(), n


def test_oracle_decodes_corpus_like_reference(orc, corpus):
    g, names = corpus
    for n in names:
        r = orc.decode(g[n + "__mp3"].tobytes())
        assert r["rc"] == 0, n
        assert np.array_equal(r["is"], g[n + "__is"]), n
        assert np.array_equal(r["bits"], g[n + "__bits"]), n
        assert r["pcm"].shape == (int(g[n + "__pcm_rows"]), int(g[n + "__nch"]))
        assert np.array_equal(r["pcm"][:2 * 1152], g[n + "__pcm_head"]), n
        assert sha(np.ascontiguousarray(r["pcm"]).tobytes()) == bytes(g[n + "__pcm_sha256"]).decode(), n
        i16 = orc.pcm_to_i16(r["pcm"])
        assert sha(i16.tobytes()) == bytes(g[n + "__pcm_i16_sha256"]).decode(), n     # wrap-around, no clipping (D13)
        assert sha(orc.wav_bytes(i16, r["sampling_rate"])) == bytes(g[n + "__wav_sha256"]).decode(), n
        assert np.array_equal(r["frames"]["main_data_begin"], g[n + "__main_data_begin"])
        assert r["bit_rate"] // 1000 == int(g[n + "__kbps"])


def test_host_parser_on_corpus(mlib, corpus):
    g, names = corpus
    for n in names:
        data = g[n + "__mp3"].tobytes()
        p = mlib.parse_stream(data)
        assert np.array_equal(p["is"], g[n + "__is"]), n
        assert np.array_equal(p["bits"], g[n + "__bits"]), n
        assert np.array_equal(p["frame_size"], g[n + "__frame_size"]), n
        nch = p["channels"]
        assert np.array_equal(p["si"]["scale_fac_l"][:, :, :nch], g[n + "__scale_fac_l"][:, :, :nch]), n
        assert np.array_equal(p["si"]["scale_fac_s"][:, :, :nch], g[n + "__scale_fac_s"][:, :, :nch]), n
        s = mlib.scan_stream(data)
        # streams with mixed blocks inherit scalefactors across frames: the scan sends them to the host parser
        assert s["gpu_ok"] == (not g[n + "__si_mixed_block_flag"].any()), n
        assert np.array_equal(s["bits"], p["bits"]) and s["n_frames"] == p["n_frames"]


@pytest.mark.gpu
def test_device_decode_of_corpus(ctx, mlib, corpus):
    g, names = corpus
    for n in names:
        data = g[n + "__mp3"].tobytes()
        r = ctx.decode_stream(data, mlib.MP3S_PCM_F64)
        assert np.array_equal(r["bits"], g[n + "__bits"]), n
        assert r["pcm"].shape == (int(g[n + "__pcm_rows"]), int(g[n + "__nch"]))
        assert sha(r["pcm"].tobytes()) == bytes(g[n + "__pcm_sha256"]).decode(), n
        r16 = ctx.decode_stream(data, mlib.MP3S_PCM_I16)
        assert sha(r16["pcm"].tobytes()) == bytes(g[n + "__pcm_i16_sha256"]).decode(), n


@pytest.mark.gpu
def test_device_huffman_on_corpus(ctx, mlib, corpus):
    """the Huffman kernel alone, also on the streams the pipeline would send to the host parser: `is` must match
    everywhere (only inherited scalefactors can differ there)"""
    L = mlib.lib()
    g, names = corpus
    for n in names:
        data = g[n + "__mp3"].tobytes()
        s = mlib.scan_stream(data)
        nf, nch = s["n_frames"], s["channels"]
        d_blob, d_side = ctx.to_device(s["blob"]), ctx.to_device(s["side"])
        d_is, d_si, d_st = ctx.alloc(nf * 2304 * 2), ctx.alloc(nf * 4 * 72), ctx.alloc(4)
        mlib.check(L.mp3s_huffman_decode_dev(ctx.handle, d_blob, d_side, nf, nch, s["max_part2_3_length"], d_is, d_si, d_st))
        ctx.sync()
        assert int(ctx.download(d_st, np.int32, (1,))[0]) == 0
        isv = ctx.download(d_is, np.int16, (nf, 2, 2, 576))
        assert np.array_equal(isv[:, :, :nch], g[n + "__is"][:, :, :nch]), n
        # every entry the requantiser reads -- long scalefactors of non-short granules, short ones of short granules --
        # including those a granule inherits from earlier frames (mixed blocks, scfsi behind a short granule 0: the kernel
        # walks back through the stream for them); and every other field of the record
        si = ctx.download(d_si, mlib.GRANULE_SI_DTYPE, (nf, 2, 2))
        p = mlib.parse_stream(data)
        bt2 = (p["si"]["block_type"] == 2)[:, :, :nch]
        assert np.array_equal(si["scale_fac_l"][:, :, :nch][~bt2], p["si"]["scale_fac_l"][:, :, :nch][~bt2]), n
        assert np.array_equal(si["scale_fac_s"][:, :, :nch][bt2], p["si"]["scale_fac_s"][:, :, :nch][bt2]), n
        # a start / stop block with the mixed flag is requantised with the SHORT scalefactors from band 8 on (Frame.py:186)
        mx = ((p["si"]["mixed_block_flag"] != 0) & (p["si"]["block_type"] != 2))[:, :, :nch]
        assert np.array_equal(si["scale_fac_s"][:, :, :nch][mx][..., 8:12], p["si"]["scale_fac_s"][:, :, :nch][mx][..., 8:12]), n
        for k in ("global_gain", "scalefac_scale", "block_type", "mixed_block_flag", "preflag", "sub_block_gain"):
            assert np.array_equal(si[k][:, :, :nch], p["si"][k][:, :, :nch]), (n, k)
        for q in (d_blob, d_side, d_is, d_si, d_st):
            ctx.free(q)


@pytest.mark.gpu
def test_device_decode_of_many_streams_in_one_batch(ctx, mlib, corpus, golden_dir):
    """a corpus of files as ONE device batch (mono and stereo groups, host-parsed and device-parsed
    streams side by side) must give, per file, exactly what the single-stream call gives = the reference's decode."""
    g, names = corpus
    files = [g[n + "__mp3"].tobytes() for n in names]
    with open(os.path.join(golden_dir, "test.mp3"), "rb") as fh:
        files.append(fh.read())
    order = list(range(len(files))) + [len(files) - 1, 0, 2]          # a file may appear more than once
    for fmt, key in ((mlib.MP3S_PCM_F64, "__pcm_sha256"), (mlib.MP3S_PCM_I16, "__pcm_i16_sha256")):
        out = ctx.decode_streams([files[i] for i in order], fmt)
        assert len(out) == len(order)
        for i, r in zip(order, out):
            single = ctx.decode_stream(files[i], fmt)
            assert r["n_frames"] == single["n_frames"] and r["channels"] == single["channels"]
            assert r["sampling_rate"] == single["sampling_rate"] and r["bit_rate"] == single["bit_rate"]
            assert np.array_equal(r["bits"], single["bits"])
            assert np.array_equal(r["pcm"], single["pcm"]), i
            if i < len(names):
                assert sha(r["pcm"].tobytes()) == bytes(g[names[i] + key]).decode(), names[i]
    w = np.load(os.path.join(golden_dir, "g3_testmp3_wav_pcm.npz"))       # the reference's own WAV of test.mp3
    assert np.array_equal(out[len(names)]["pcm"], w["pcm"])


@pytest.mark.gpu
def test_many_streams_rejects_a_malformed_member(ctx, mlib, corpus):
    g, names = corpus
    good = g[names[0] + "__mp3"].tobytes()
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.decode_streams([good, b"\xff\xfb\x90", good])              # a sync with no header behind it
    assert "file 1" in str(e.value)
    assert ctx.decode_streams([]) == []
    # no sync where the stream should start: the reference parses nothing and writes an empty WAV; here: an empty member
    out = ctx.decode_streams([good, b"\x00" * 64, good])
    assert out[1]["n_frames"] == 0 and out[1]["pcm"].size == 0 and out[1]["bit_rate"] == 0
    assert np.array_equal(out[0]["pcm"], out[2]["pcm"]) and out[0]["n_frames"] > 0


@pytest.mark.gpu
def test_many_streams_per_file_status(ctx, mlib, corpus):
    """per_file=True: every member gets its own status (what the single-file call fails with), the others decode as in a
    batch of their own -- the rule mp3s_hide_messages already follows"""
    g, names = corpus
    files = [g[n + "__mp3"].tobytes() for n in names]
    bad = [b"\xff\xfb\x90", b"", b"\xff\xfb\x90\x64" + b"\x00" * 10]
    mixed = [files[0], bad[0], files[1], bad[1], files[2], bad[2]] + files[3:]
    out = ctx.decode_streams(mixed, mlib.MP3S_PCM_F64, per_file=True)
    assert len(out) == len(mixed)
    for i, (f, r) in enumerate(zip(mixed, out)):
        try:
            single = ctx.decode_stream(f, mlib.MP3S_PCM_F64)
        except mlib.Mp3sError as e:
            assert isinstance(r, mlib.Mp3sError) and r.code == e.code, (i, r, e)
            continue
        assert not isinstance(r, Exception), (i, r)
        assert r["pcm"].tobytes() == single["pcm"].tobytes() and np.array_equal(r["bits"], single["bits"]), i
    assert sum(isinstance(r, Exception) for r in out) >= 2


@pytest.mark.gpu
def test_inherited_scalefactors_far_back_and_in_a_batch(ctx, mlib, orc):
    """long streams full of mixed blocks and short / long switches: the entries a granule inherits were written many frames
    earlier; several such streams in one batch (the walk stops at the stream's own first frame); against the host parser's
    samples and against the oracle's PCM"""
    import frame_synth
    files = [frame_synth.make_stream(20 + i, 90 + 17 * i, block_types=(0, 1, 2, 3), allow_mixed=True, mode=(0, 1, 3)[i % 3],
                                     mode_ext=2 if i % 3 == 1 else 0) for i in range(5)]
    files.append(frame_synth.make_stream(40, 60, block_types=(0,)))             # a plain one in between
    assert sum(not mlib.scan_stream(f)["gpu_ok"] for f in files) >= 4
    for f in files:
        o = orc.decode(f)
        got = ctx.decode_stream(f, mlib.MP3S_PCM_F64)
        assert got["pcm"].tobytes() == o["pcm"].tobytes()
    order = [3, 0, 5, 1, 4, 2]
    batch = ctx.decode_streams([files[i] for i in order], mlib.MP3S_PCM_F64)
    for i, r in zip(order, batch):
        assert r["pcm"].tobytes() == orc.decode(files[i])["pcm"].tobytes(), i
    # blocks of such a stream still go through the host parser: same samples
    f = files[0]
    whole = ctx.decode_stream(f, mlib.MP3S_PCM_F64)["pcm"]
    blk = ctx.decode_block(f, 40, 30, mlib.MP3S_PCM_F64)
    assert blk["pcm"].tobytes() == whole[40 * 1152:70 * 1152].tobytes()
