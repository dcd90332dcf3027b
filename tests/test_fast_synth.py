"""The fast int16 decode (k_dec_stream; k_dec_imdct<true> + k_dec_synth_fast with MP3S_OPT_FUSED_DECODE = 0) behind its guard: int16 PCM equal to the exact kernel's -- which is the
reference's, bit for bit (tests/test_gpu_parity.py, tests/test_decode_corpus.py) -- also when the guard is made so wide that
most samples take the exact path, and on streams that start and stop inside a tile."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fast_synthesis_equals_exact_kernel(mlib, golden_dir, orc):
    from synth_pcm import synth_pcm
    import frame_synth
    ctx = mlib.Context(0)
    try:
        rng = np.random.default_rng(8)
        pcm = synth_pcm(400, seed=51)
        pcm[100 * 1152:120 * 1152] = 0
        loud = (rng.integers(-32768, 32767, size=(200 * 1152, 2))).astype(np.int16)          # full-scale noise: large |S|
        streams = [bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"]), bytes(ctx.encode_pcm(loud, 48000, 320, None)["mp3"]),
                   frame_synth.make_stream(31, 150, block_types=(0, 1, 2, 3), allow_mixed=True, mode=1, mode_ext=2),
                   frame_synth.make_stream(32, 90, mode=3, max_lin=8191),                      # mono, values up to +-8206
                   open(os.path.join(golden_dir, "test.mp3"), "rb").read()]
        ctx.synth_mode(0.0)
        exact = [np.array(ctx.decode_stream(s, mlib.MP3S_PCM_I16)["pcm"]) for s in streams]
        exact_batch = [np.array(r["pcm"]) for r in ctx.decode_streams(streams, mlib.MP3S_PCM_I16)]
        assert np.array_equal(exact[4], orc.pcm_to_i16(orc.decode(streams[4])["pcm"]))
        # both fast paths: the one-kernel stream (k_dec_stream, the default) and the two kernels with S in device memory
        for fused in (0, 1):
            ctx.set_option("fused_decode", fused)
            counts = {}
            for scale in (1.0, 1e3, 1e6, 1e9):
                ctx.synth_mode(scale)
                fast = [np.array(ctx.decode_stream(s, mlib.MP3S_PCM_I16)["pcm"]) for s in streams]
                batch = [np.array(r["pcm"]) for r in ctx.decode_streams(streams, mlib.MP3S_PCM_I16)]
                counts[scale] = ctx.synth_mode(scale)
                for k in range(len(streams)):
                    assert np.array_equal(fast[k], exact[k]), (fused, scale, k)
                    assert np.array_equal(batch[k], exact_batch[k]), (fused, scale, k)
        # the guard's share: next to nothing at the proven bound (a flagged sample takes the 32 of its slot and channel with it),
        # a large share when inflated a billion times
        total = 2 * sum(e.size for e in exact)
        assert counts[1.0] < total * 1e-3 and counts[1e9] > total * 0.5, counts
        assert counts[1.0] <= counts[1e3] <= counts[1e6] <= counts[1e9]
        # the float formats do not take the fast path
        ctx.synth_mode(1.0)
        f64 = ctx.decode_stream(streams[4], mlib.MP3S_PCM_F64)["pcm"]
        assert f64.tobytes() == orc.decode(streams[4])["pcm"].tobytes()
    finally:
        ctx.close()


def test_tiles_that_could_leave_int32_go_to_the_exact_order_as_a_whole(mlib):
    """k_dec_synth_fast vouches for a tile's truncations only when synth_xbound * (largest sum |S| of a slot) stays inside int32
    (its per-sample conversion saturates where the reference's wraps, MP3_Parser.py:91); a tile beyond that is put on the
    fix-up list whole.  A mono stream with escape values up to 8 206: samples far beyond int32, int16 output equal to the
    exact kernel's, and a large share of the samples recomputed at the guard's own width"""
    import frame_synth
    ctx = mlib.Context(0)
    try:
        mp3 = bytearray(frame_synth.make_stream(32, 90, mode=3, max_lin=8191))
        # global_gain = 255 in every granule (mono side info: 9 + 5 + 4 bits, then per granule 12 + 9 in front of its 8 bits, 59 in all)
        pos = 0
        for size in mlib.parse_stream(bytes(mp3))["frame_size"]:
            for gr in range(2):
                bit = (pos + 4) * 8 + 18 + 59 * gr + 21
                for b in range(bit, bit + 8):
                    mp3[b >> 3] |= 0x80 >> (b & 7)
            pos += int(size)
        mp3 = bytes(mp3)
        f64 = np.array(ctx.decode_stream(mp3, mlib.MP3S_PCM_F64)["pcm"])
        beyond = int(np.count_nonzero(np.abs(f64) * 32767 >= 2147483648.0))
        assert beyond > 0, "the stream no longer reaches beyond int32: pick another seed"
        ctx.synth_mode(0.0)
        exact = np.array(ctx.decode_stream(mp3, mlib.MP3S_PCM_I16)["pcm"])
        ctx.synth_mode(1.0)
        fast = np.array(ctx.decode_stream(mp3, mlib.MP3S_PCM_I16)["pcm"])
        n_exact = ctx.synth_mode(1.0)
        assert np.array_equal(fast, exact)
        assert n_exact >= beyond and n_exact > exact.size // 20, (n_exact, beyond, exact.size)
    finally:
        ctx.close()


def test_fast_decode_across_transform_chunks(mlib):
    """A stream longer than one transform chunk (16 384 frames, one frame of halo in front of the second chunk) with the
    guard inflated: the fix-up kernel recomputes flagged samples of both chunks from `is`, with the halo frame as priming."""
    from synth_pcm import synth_pcm
    ctx = mlib.Context(0)
    try:
        pcm = synth_pcm(17000, seed=77)
        mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
        ctx.synth_mode(0.0)
        exact = np.array(ctx.decode_stream(mp3, mlib.MP3S_PCM_I16)["pcm"])
        for scale in (1.0, 1e5):
            ctx.synth_mode(scale)
            fast = np.array(ctx.decode_stream(mp3, mlib.MP3S_PCM_I16)["pcm"])
            n_exact = ctx.synth_mode(1.0)
            assert np.array_equal(fast, exact), scale
            if scale > 1:
                assert n_exact > 100000, n_exact          # a flood through the fix-up kernel, second chunk included
        del exact, fast
    finally:
        ctx.close()


@pytest.mark.gpu
def test_float32_through_the_fast_sums_is_an_option_within_the_contract(ctx, mlib, orc, golden_dir):
    """MP3S_OPT_FLOAT_FAST: float32 output through the mirrored, fused IMDCT and the split synthesis, without guard or fix-up
    (decoder/Frame.py:65-154).  Off (the default) the float formats stay bit-identical to the reference; on, every sample is within
    the contract's 1e-5 relative tolerance (measured: a few float32 last places) on the reference's own file, the decode corpus
    (short / mixed blocks, MS, mono, reservoir), a stream that starts inside tiles, and the chunked calls"""
    import frame_synth
    from synth_pcm import synth_pcm
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    streams = [open(os.path.join(golden_dir, "test.mp3"), "rb").read()]
    streams += [g[k].tobytes() for k in g.files if k.endswith("__mp3")]
    streams.append(bytes(ctx.encode_pcm(synth_pcm(700, seed=81), 44100, 128, None)["mp3"]))
    streams.append(frame_synth.make_stream(61, 300, block_types=(0, 1, 2, 3), allow_mixed=True, use_reservoir=True))
    worst_rel, worst_abs, differing = 0.0, 0.0, 0
    keep = ctx.get_option("float_fast")
    try:
        for data in streams:
            ref64 = orc.decode(data)["pcm"]
            ctx.set_option("float_fast", 0)
            exact = ctx.decode_stream(data, mlib.MP3S_PCM_F32)["pcm"]
            assert np.array_equal(exact, ref64.astype(np.float32))                       # the default: the reference's samples, rounded once
            ctx.set_option("float_fast", 1)
            for chunk in (0, 37):
                ctx.set_option("chunk_frames", chunk)
                fast = ctx.decode_stream(data, mlib.MP3S_PCM_F32)["pcm"]
                assert fast.shape == exact.shape
                assert np.allclose(fast, ref64, rtol=1e-5, atol=1e-9), len(data)
                d = np.abs(fast.astype(np.float64) - ref64)
                big = np.abs(ref64) > 1e-6
                worst_abs = max(worst_abs, float(d.max()) if d.size else 0.0)
                if big.any():
                    worst_rel = max(worst_rel, float((d[big] / np.abs(ref64[big])).max()))
                differing += int((fast != exact).sum())
            ctx.set_option("chunk_frames", 0)
            # the other formats do not change with the option
            assert ctx.decode_stream(data, mlib.MP3S_PCM_F64)["pcm"].tobytes() == ref64.tobytes()
            assert np.array_equal(ctx.decode_stream(data, mlib.MP3S_PCM_I16)["pcm"], orc.pcm_to_i16(ref64))
    finally:
        ctx.set_option("float_fast", keep)
        ctx.set_option("chunk_frames", 0)
    assert worst_rel < 1e-6, (worst_rel, worst_abs)          # (measured 6e-8: float32's own rounding; the corpus holds samples up to +-1340)
    assert differing > 0            # (the fast sums are not the reference's: if nothing differs the option did not take)
