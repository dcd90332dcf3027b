"""Parity tests proper: the HIP path (through the C-ABI) against the oracle and the reference goldens.
Bit-exact everywhere: integer artefacts, MP3 bytes, stego bits AND the float64 PCM (the kernels keep the
reference's fp64 operation order), so the 1e-5 relative tolerance of the contract is met with margin 0."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def bits_of(s):
    return np.frombuffer("".join(format(b, "08b") for b in s.encode()).encode(), dtype=np.uint8) - ord("0")


# ------------------------------------------------------------------------------------------------ decode
def _chain_batch(mlib, g, name):
    hdr_b, isv = g[name + "__hdr"], g[name + "__is"]
    n = isv.shape[0]
    mode = int(hdr_b[3] >> 6)
    nch = 1 if mode == 3 else 2
    si = np.zeros((n, 2, 2), dtype=mlib.GRANULE_SI_DTYPE)
    si["global_gain"] = g[name + "__gg"]
    si["scalefac_scale"] = g[name + "__sfs"]
    si["block_type"] = g[name + "__bt"]
    si["mixed_block_flag"] = g[name + "__mixed"]
    si["preflag"] = g[name + "__pre"]
    si["sub_block_gain"] = g[name + "__sbg"]
    si["scale_fac_l"] = g[name + "__sfl"]
    si["scale_fac_s"] = g[name + "__sfsh"]
    hdr = np.zeros(n, dtype=mlib.FRAME_HDR_DTYPE)
    hdr["sr_idx"] = (hdr_b[2] >> 2) & 3
    hdr["nch"] = nch
    hdr["ms_stereo"] = 1 if (mode == 1 and (hdr_b[3] & 0x20)) else 0
    return isv, si, hdr, nch


def test_decode_transform_reference_chain(ctx, mlib, golden_dir):
    """short / start / stop / mixed blocks, MS stereo, mono, 32/44.1/48 kHz: float64 PCM bit for bit"""
    g = np.load(os.path.join(golden_dir, "g4_decode_chain.npz"))
    for name in sorted({k.split("__")[0] for k in g.files}):
        isv, si, hdr, nch = _chain_batch(mlib, g, name)
        pcm = ctx.decode_transform(isv, si, hdr, nch, 0, mlib.MP3S_PCM_F64)
        ref = g[name + "__pcm"].reshape(-1, nch)
        assert pcm.shape == ref.shape
        assert pcm.tobytes() == ref.tobytes(), name
        f32 = ctx.decode_transform(isv, si, hdr, nch, 0, mlib.MP3S_PCM_F32)
        assert np.array_equal(f32, ref.astype(np.float32)), name


def test_decode_batch_of_streams_and_halo(ctx, mlib, orc, golden_dir):
    """several streams in one batch (stream_first) and chunked decoding with a 1-frame halo"""
    g = np.load(os.path.join(golden_dir, "g4_decode_chain.npz"))
    a = _chain_batch(mlib, g, "all_44_ms")
    b = _chain_batch(mlib, g, "long_44_stereo")
    ra = g["all_44_ms__pcm"].reshape(-1, 2)
    rb = g["long_44_stereo__pcm"].reshape(-1, 2)
    isv = np.concatenate([a[0], b[0], a[0]])
    si = np.concatenate([a[1], b[1], a[1]])
    hdr = np.concatenate([a[2], b[2], a[2]])
    na, nb = len(a[2]), len(b[2])
    hdr["stream_first"] = [0] * na + [na] * nb + [na + nb] * na
    pcm = ctx.decode_transform(isv, si, hdr, 2, 0, mlib.MP3S_PCM_F64)
    assert pcm.tobytes() == np.concatenate([ra, rb, ra]).tobytes()
    # chunked decode: a chunk that starts at frame `start` needs only frame start-1 as halo -- the IMDCT overlap
    # reaches back one granule and the V fifo 15 slots, both inside the halo frame's second granule
    p = mlib.parse_stream(open(os.path.join(golden_dir, "test.mp3"), "rb").read())
    full = ctx.decode_transform(p["is"], p["si"], p["hdr"], 2, 0, mlib.MP3S_PCM_F64)
    for start in (1, 2, 7, 20, 35):
        part = ctx.decode_transform(p["is"][start - 1:], p["si"][start - 1:], p["hdr"][start - 1:], 2, 1, mlib.MP3S_PCM_F64)
        assert part.tobytes() == full[start * 1152:].tobytes(), start
    i16 = ctx.decode_transform(p["is"][6:], p["si"][6:], p["hdr"][6:], 2, 1, mlib.MP3S_PCM_I16)
    assert np.array_equal(i16, orc.pcm_to_i16(full[7 * 1152:]))


def test_decode_stream_testmp3(ctx, mlib, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_decode_testmp3.npz"))
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    r = ctx.decode_stream(data, mlib.MP3S_PCM_F64)
    assert r["n_frames"] == 36 and r["bit_rate"] == 320000 and r["sampling_rate"] == 44100
    assert np.array_equal(r["bits"], g["bits"])
    assert np.array_equal(r["pcm"][:4 * 1152], g["pcm_head"])
    assert sha(r["pcm"].tobytes()) == bytes(g["pcm_sha256"]).decode()
    r16 = ctx.decode_stream(data, mlib.MP3S_PCM_I16)
    assert sha(r16["pcm"].tobytes()) == bytes(g["pcm_i16_sha256"]).decode()
    assert sha(orc.wav_bytes(r16["pcm"], 44100)) == bytes(g["wav_sha256"]).decode()


def test_pcm_i16_wraps_like_numpy(ctx, mlib, orc):
    """(pcm*32767).astype(int16): truncation toward zero and wrap-around, no clipping"""
    n = 2
    isv = np.zeros((n, 2, 2, 576), dtype=np.int16)
    isv[:, :, :, :40] = 8000
    si = np.zeros((n, 2, 2), dtype=mlib.GRANULE_SI_DTYPE)
    si["global_gain"] = 200
    hdr = np.zeros(n, dtype=mlib.FRAME_HDR_DTYPE)
    hdr["nch"] = 2
    f64 = ctx.decode_transform(isv, si, hdr, 2, 0, mlib.MP3S_PCM_F64)
    i16 = ctx.decode_transform(isv, si, hdr, 2, 0, mlib.MP3S_PCM_I16)
    assert np.abs(f64).max() > 1.5                       # far outside [-1, 1]
    assert np.array_equal(i16, orc.pcm_to_i16(f64))


# ------------------------------------------------------------------------------------------------ encode
def test_encode_transform_golden(ctx, mlib, orc, golden_dir):
    pcm = np.load(os.path.join(golden_dir, "g3_testmp3_wav_pcm.npz"))["pcm"]
    g = np.load(os.path.join(golden_dir, "g3_encode_plain320.npz"))
    mdct = ctx.encode_transform(pcm)
    assert np.array_equal(mdct[:4], g["mdct_freq"])
    assert np.array_equal(mdct, orc.encode(pcm, 44100, 320)["mdct_freq"])
    # extreme samples
    rng = np.random.default_rng(7)
    hard = rng.choice(np.array([-32768, 32767, 0, 1, -1], dtype=np.int16), size=(8 * 1152, 2))
    want_hard = orc.encode(hard, 44100, 320)["mdct_freq"]
    assert np.array_equal(ctx.encode_transform(hard), want_hard)
    # the same through the one-kernel form (MP3S_OPT_FUSED_ENCODE: analysis and MDCT in one launch, the subband samples in LDS): tiles of 13
    # granules, so 60 frames make ten workgroups whose first granule is even and odd in turn
    long_pcm = np.concatenate([pcm, hard, pcm[:16 * 1152]])
    want_long = ctx.encode_transform(long_pcm)
    assert np.array_equal(want_long, orc.encode(long_pcm, 44100, 320)["mdct_freq"])
    old = ctx.set_option("fused_encode", 1)
    try:
        assert np.array_equal(ctx.encode_transform(pcm), mdct)
        assert np.array_equal(ctx.encode_transform(hard), want_hard)
        assert np.array_equal(ctx.encode_transform(long_pcm), want_long)
    finally:
        ctx.set_option("fused_encode", old)


@pytest.mark.parametrize("name", ["plain320", "hide_ddd320"])
def test_encode_pcm_golden(ctx, mlib, golden_dir, name):
    pcm = np.load(os.path.join(golden_dir, "g3_testmp3_wav_pcm.npz"))["pcm"]
    g = np.load(os.path.join(golden_dir, f"g3_encode_{name}.npz"))
    hide = g["hide_bits"] if "hide_bits" in g.files else None
    r = ctx.encode_pcm(pcm, 44100, 320, hide)
    assert len(r["mp3"]) == int(g["mp3_len"]) and sha(r["mp3"]) == bytes(g["mp3_sha256"]).decode()
    _check_gr(r, g)
    assert r["too_long"] == bool(int(g["too_long"]))
    assert r["hide_offset"] == int(g["hide_off"][-1])


def _check_gr(r, g):
    """every table index and side-info field of every granule*channel"""
    fields = bytes(g["gi_fields"]).decode().split(",")
    gi = g["gi"]                                  # [f][gr][ch][field]
    n = gi.shape[0]
    gr = r["gr"].reshape(n, 2, 2)                 # [f][ch][gr]
    for a, b in (("big_values", "big_values"), ("count1", "count1"), ("quantizer_step", "quantizerStepSize"),
                 ("region0_count", "region0_count"), ("region1_count", "region1_count"),
                 ("count1table_select", "count1table_select")):
        assert np.array_equal(gr[a].transpose(0, 2, 1), gi[..., fields.index(b)]), a
    assert np.array_equal(gr["table_select"].transpose(0, 2, 1, 3), g["table_select"])
    for k, fld in enumerate(("address1", "address2", "address3")):
        assert np.array_equal(gr["address"][..., k].transpose(0, 2, 1), gi[..., fields.index(fld)]), fld
    assert np.array_equal(r["scfsi"], g["scfsi"])


def test_encode_synth128_with_silence(ctx, mlib, golden_dir):
    """128 kbps synthetic stream ending in digital silence: big_values == 0 units, stale address1/2/3 and
    quantizerStepSize inheritance, 56-bit message"""
    g = np.load(os.path.join(golden_dir, "g6_synth128.npz"))
    r = ctx.encode_pcm(g["pcm"], 44100, 128, g["hide_bits"])
    assert r["mp3"] == g["mp3"].tobytes()
    _check_gr(r, g)
    d = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_F64)
    assert np.array_equal(d["bits"], g["dec_bits"])
    assert np.array_equal(d["bits"][:len(g["hide_bits"])], g["hide_bits"])
    assert sha(d["pcm"].tobytes()) == bytes(g["dec_pcm_sha256"]).decode()


def test_rate_loop_batch_entry_point(ctx, mlib, orc, golden_dir):
    """mp3s_rate_loop_dev on device buffers: unit lists, inherited state, energies vs the host (glibc) evaluation"""
    import ctypes as C
    g = np.load(os.path.join(golden_dir, "g6_synth128.npz"))
    o = orc.encode(g["pcm"], 44100, 128, None)
    n = o["n_frames"]
    units = n * 4
    L = mlib.lib()
    mdct = np.ascontiguousarray(o["mdct_freq"], dtype=np.int32)
    rf, _ = mlib.rate_frames(44100, 128, 2, n)
    d_mdct, d_rf = ctx.to_device(mdct), ctx.to_device(rf)
    d_ix, d_out, d_en = ctx.alloc(units * 576 * 2), ctx.alloc(units * 72), ctx.alloc(units * 22 * 4)
    d_state = ctx.to_device(np.zeros((units, 4), dtype=np.int32))
    mlib.check(L.mp3s_rate_loop_dev(ctx.handle, d_mdct, d_rf, n, None, 0, None, d_state, None, 0, d_ix, d_out, d_en))
    ctx.sync()
    out = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
    en = ctx.download(d_en, np.int32, (units, 22))
    ix = ctx.download(d_ix, np.int16, (n, 2, 2, 576))
    act = (out["flags"] & mlib.RF_ACTIVE) != 0
    gi = o["frames"]["gi"]
    ref_bv = gi["big_values"].transpose(0, 2, 1).reshape(-1)
    ref_ts = gi["table_select"].transpose(0, 2, 1, 3).reshape(-1, 3)
    # units that did not inherit state are final after one launch
    clean = act & ((out["flags"] & mlib.RF_USED_ADDR_IN) == 0)
    assert clean.sum() > 100
    assert np.array_equal(out["big_values"][clean], ref_bv[clean])
    assert np.array_equal(out["table_select"][clean], ref_ts[clean])
    assert np.array_equal(ix.reshape(units, 576)[clean], o["ix"].reshape(units, 576)[clean].astype(np.int16))
    # device log vs glibc log: identical truncated energies (and the guard flag never fired here)
    assert not (out["flags"] & mlib.RF_LOG_GUARD).any()
    host_en = np.zeros(22, dtype=np.int32)
    for u in list(range(0, units, 7)) + [units - 1]:
        xr = np.ascontiguousarray(mdct.reshape(units, 576)[u])
        mlib.check(L.mp3s_debug_scfsi_energies(xr.ctypes.data, 0, host_en.ctypes.data))
        assert np.array_equal(en[u], host_en), u
    # a unit list recomputes only the listed units
    lst = np.array([5, 17, 40], dtype=np.int32)
    mlib.check(L.mp3s_dev_memset(ctx.handle, d_out, 0, units * 72))
    d_list = ctx.to_device(lst)
    mlib.check(L.mp3s_rate_loop_dev(ctx.handle, d_mdct, d_rf, n, None, 0, None, d_state, d_list, 3, d_ix, d_out, d_en))
    ctx.sync()
    out2 = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
    for u in range(units):
        if u in lst:
            assert out2[u] == out[u]
        else:
            assert out2[u]["flags"] == 0 and out2[u]["xrmax"] == 0
    for p in (d_mdct, d_rf, d_ix, d_out, d_en, d_state, d_list):
        ctx.free(p)


@pytest.mark.parametrize("rate,kbps", [(44100, 32), (44100, 48), (48000, 56), (32000, 32), (44100, 128), (48000, 320)])
def test_rate_loop_on_sparse_spectra(ctx, mlib, orc, rate, kbps):
    """mp3s_rate_loop_dev on spectra with empty regions below the last big value, lone lines, everything at the first thresholds
    (tests/spectra.py), at budgets from 32 to 320 kbit/s: every unit's step, run lengths, regions, tables, bits and quantised lines
    against the reference's loop (oracle rate_units: MP3_Encoder.py:766-813) -- the binary search's decisions, which the pre-check
    and the bounds take without evaluating a probe in full, must be the reference's everywhere (round-5 advisor finding:
    tests/test_rate_bounds.py has the bound itself)."""
    from spectra import sparse_spectra
    t = mlib.debug_tables()
    n = 1024                                                          # frames: 4096 spectra
    units = 4 * n
    parts = [sparse_spectra(rate + kbps + 7 * k, units // 4, base=max(int(t["rl_t1"][st + 127]), 1))
             for k, st in enumerate((-60, -60, -30, -90))]           # aimed at the first probe and at both second probes
    xr = np.ascontiguousarray(np.concatenate(parts)[np.random.default_rng(kbps).permutation(units)])
    rf, _ = mlib.rate_frames(rate, kbps, 2, n)
    want = orc.rate_units(rate, np.repeat(rf["max_bits"], 4), xr)
    L = mlib.lib()
    d_mdct, d_rf = ctx.to_device(xr), ctx.to_device(rf)
    d_ix, d_out, d_en = ctx.alloc(units * 576 * 2), ctx.alloc(units * 72), ctx.alloc(units * 22 * 4)
    d_state = ctx.to_device(np.zeros((units, 4), dtype=np.int32))
    try:
        mlib.check(L.mp3s_rate_loop_dev(ctx.handle, d_mdct, d_rf, n, None, 0, None, d_state, None, 0, d_ix, d_out, d_en))
        ctx.sync()
        out = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
        ix = ctx.download(d_ix, np.int16, (units, 576)).astype(np.int32)
    finally:
        for p in (d_mdct, d_rf, d_ix, d_out, d_en, d_state):
            ctx.free(p)
    gi = want["gi"]
    ok = want["rc"] == 0
    assert ok.sum() > units * 0.9
    assert np.array_equal(((out["flags"] & mlib.RF_STEP_RANGE) != 0), ~ok)     # the quantiser step ran out of steptab: IndexError in the reference
    act = (np.abs(xr).max(1) > 0) & ok
    assert np.array_equal((out["flags"] & mlib.RF_ACTIVE) != 0, np.abs(xr).max(1) > 0)
    for a, b in (("quantizer_step", "quantizerStepSize"), ("big_values", "big_values"), ("count1", "count1"),
                 ("part2_3_length", "part2_3_length"), ("region0_count", "region0_count"), ("region1_count", "region1_count"),
                 ("count1table_select", "count1table_select"), ("table_select", "table_select")):
        bad = np.nonzero(act & ~(out[a] == gi[b]).reshape(units, -1).all(1))[0]
        assert bad.size == 0, (a, bad[:8], out[a][bad[:4]], gi[b][bad[:4]])
    for k, fld in enumerate(("address1", "address2", "address3")):
        assert np.array_equal(out["address"][act, k], gi[fld][act]), fld
    assert np.array_equal(np.abs(ix[act]), want["ix"][act])
    assert ((ix[act] == 0) | ((ix[act] < 0) == (xr[act] < 0))).all()


def test_facade_hashes(ctx, mlib, golden_dir):
    """hide / clear / too-long through the device pipeline equal the reference facade run"""
    fac = json.load(open(os.path.join(golden_dir, "g3_facade.json")))
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    d = ctx.decode_stream(data, mlib.MP3S_PCM_I16)
    kbps = d["bit_rate"] // 1000
    h = ctx.encode_pcm(d["pcm"], d["sampling_rate"], kbps, bits_of("3#ddd"))
    assert sha(h["mp3"]) == fac["hide_sha256"] and h["too_long"] == fac["too_long"]
    d2 = ctx.decode_stream(h["mp3"], mlib.MP3S_PCM_I16)
    c = ctx.encode_pcm(d2["pcm"], d2["sampling_rate"], d2["bit_rate"] // 1000, None)
    assert sha(c["mp3"]) == fac["cleared_sha256"]
    lg = ctx.encode_pcm(d["pcm"], d["sampling_rate"], kbps, bits_of("300#" + "ddd" * 100))
    assert sha(lg["mp3"]) == fac["hide_long_sha256"] and lg["too_long"] == fac["too_long_300"]


def test_encode_errors_like_reference(ctx, mlib):
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.encode_pcm(np.zeros((1152, 1), dtype=np.int16), 44100, 128)
    assert e.value.code == mlib.E_UNSUPPORTED
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.encode_pcm(np.zeros((1152 + 7, 2), dtype=np.int16), 44100, 128)
    assert e.value.code == mlib.E_UNSUPPORTED
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.encode_pcm(np.zeros((1152, 2), dtype=np.int16), 22050, 128)
    assert e.value.code == mlib.E_UNSUPPORTED
    # all-zero input: every unit silent, nothing hidden
    r = ctx.encode_pcm(np.zeros((3 * 1152, 2), dtype=np.int16), 44100, 128, bits_of("1#x"))
    assert r["hide_offset"] == 0 and r["too_long"] and (r["gr"]["big_values"] == 0).all()


# ------------------------------------------------------------------------------------------------ full size
@pytest.mark.parametrize("rate,kbps", [(44100, 128), (48000, 320), (32000, 64)])
def test_full_size_against_oracle(ctx, mlib, orc, rate, kbps):
    """BASELINE-size batch (10k frames at 44.1k/128k; 2k at the other rates): the whole pipeline against
    the oracle, byte for byte, plus the size-independent round-trip property"""
    from synth_pcm import synth_pcm
    n = 10000 if rate == 44100 else 2000
    pcm = synth_pcm(n, rate=rate)
    msg = bits_of("64#" + "The quick brown fox jumps over the lazy dog, again & again, 0123")
    r = ctx.encode_pcm(pcm, rate, kbps, msg)
    o = orc.encode(pcm, rate, kbps, msg)
    assert o["rc"] == 0 and r["mp3"] == o["mp3"]
    assert r["hide_offset"] == o["hide_offset"] and r["too_long"] == bool(o["too_long"])
    d = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_F64)
    od = orc.decode(r["mp3"])
    assert d["n_frames"] == od["n_frames"]
    assert np.array_equal(d["bits"], od["bits"])
    assert np.array_equal(d["bits"][:len(msg)], msg)                 # encode -> decode recovers the message
    assert sha(d["pcm"].tobytes()) == sha(np.ascontiguousarray(od["pcm"]).tobytes())
    d16 = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_I16)
    assert np.array_equal(d16["pcm"], orc.pcm_to_i16(od["pcm"]))
    # decode -> re-encode (clear) -> decode: no message left, same number of frames
    c = ctx.encode_pcm(d16["pcm"], rate, kbps, None)
    assert c["mp3"] == orc.encode(d16["pcm"], rate, kbps, None)["mp3"]


# ------------------------------------------------------------------------------------------------ BASELINE config 5
@pytest.mark.parametrize("rate", [32000, 44100, 48000])
def test_encode_matrix_of_rates_and_bitrates(ctx, mlib, orc, rate):
    """every (sampling rate, bitrate) of BASELINE config 5 in stereo: bytes, cursor and the decode of the result
    against the oracle (mono encode is a reference crash, test_encode_errors_like_reference)"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(160, rate=rate, seed=0x1234 + rate)
    pcm[40 * 1152:44 * 1152] = 0                                      # a silent stretch: empty units, reservoir refill
    pcm[100 * 1152:101 * 1152, 0] = 32767                             # a clipped burst: the quantiser's float path
    msg = bits_of("20#" + "rates and bitrates..")
    for kbps in (32, 64, 128, 192, 320):
        r = ctx.encode_pcm(pcm, rate, kbps, msg)
        o = orc.encode(pcm, rate, kbps, msg)
        assert o["rc"] == 0 and r["mp3"] == o["mp3"], (rate, kbps)
        assert r["hide_offset"] == o["hide_offset"] and r["too_long"] == bool(o["too_long"])
        d = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_F64)
        od = orc.decode(r["mp3"])
        assert d["bit_rate"] == od["bit_rate"] == kbps * 1000 and d["sampling_rate"] == rate
        assert np.array_equal(d["bits"], od["bits"])
        assert np.array_equal(d["pcm"], od["pcm"]), (rate, kbps)


def test_pilot_tone_in_digital_silence_at_low_bit_rates(ctx, mlib, orc):
    """The round-5 advisor's case as PCM: a low-level high-frequency tone in digital silence at 32 / 48 / 56 kbit/s -- big_values reaches far up,
    every region below the tone is empty (table 0: free in the reference, encoder/MP3_Encoder.py:1182-1184, :228-229), and the binary search's first
    probe has FEWER bits than max_bits although a bit per pair below big_values would say otherwise (the premise is checked on the oracle's own
    spectra: tests/test_rate_bounds.py's model of the round-5 bound).  Bytes, cursor and every granule's side info against the oracle, with and
    without a message; the decode of the result as well."""
    from test_rate_bounds import precheck_model
    t = mlib.debug_tables()
    t1, t2 = int(t["rl_t1"][67]), int(t["rl_t2"][67])
    n = 48
    misfires = 0
    for rate, freq, amp, kbps in ((44100, 19000, 5, 48), (44100, 15000, 2, 32), (44100, 20500, 20, 56), (48000, 21000, 8, 48), (32000, 14500, 6, 48)):
        tt = np.arange(n * 1152) / rate
        sig = np.rint(amp * np.sin(2 * np.pi * freq * tt)).astype(np.int16)
        pcm = np.ascontiguousarray(np.stack([sig, np.roll(sig, 7)], axis=1))
        pcm[20 * 1152:24 * 1152] = 0                                  # silent frames inside: inherited addresses
        for msg in (None, bits_of("12#pilot tones..")):
            o = orc.encode(pcm, rate, kbps, msg)
            r = ctx.encode_pcm(pcm, rate, kbps, msg)
            assert o["rc"] == 0 and r["mp3"] == o["mp3"], (rate, freq, amp, kbps, msg is not None)
            assert r["hide_offset"] == o["hide_offset"]
            gi = o["frames"]["gi"]
            gr = r["gr"].reshape(-1, 2, 2)
            assert np.array_equal(gr["quantizer_step"].transpose(0, 2, 1), gi["quantizerStepSize"])
            assert np.array_equal(gr["big_values"].transpose(0, 2, 1), gi["big_values"])
            assert np.array_equal(gr["table_select"].transpose(0, 2, 1, 3), gi["table_select"])
        d, od = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_F64), orc.decode(r["mp3"])
        assert np.array_equal(d["pcm"], od["pcm"]) and np.array_equal(d["bits"], od["bits"])
        # the premise: units whose first probe the round-5 bound would have decided the other way
        md = o["mdct_freq"].reshape(-1, 576)
        rf, _ = mlib.rate_frames(rate, kbps, 2, n)
        mb = np.repeat(rf["max_bits"], 4)
        bits, _, _ = orc.probe_bits(rate, -60, md)
        _, _, lb, lb_r5 = precheck_model(md, t1, t2)
        ran = (bits >= 0) & (bits < 100000)
        assert (lb[ran] <= bits[ran]).all()
        misfires += int((ran & (bits < mb) & (lb_r5 >= mb)).sum())
    assert misfires > 100, misfires


# ------------------------------------------------------------------------------------------------ BASELINE config 4 (shape)
def test_many_seeded_streams_across_chunk_boundaries(ctx, mlib, orc):
    """config 4 in miniature: several seeded streams as one batch that is larger than the pipeline's transform chunk,
    so chunk boundaries fall inside streams (halo re-run) and stream starts fall inside chunks (state reset);
    per stream a CRC of the int16 PCM against the oracle, and the float64 PCM exactly on one of them"""
    import zlib
    from synth_pcm import synth_pcm
    lens = [3000, 5000, 1, 4200, 2, 6100, 700]                        # 19003 frames > 16384
    files = []
    for i, n in enumerate(lens):
        pcm = synth_pcm(n, seed=0x9E3779B97F4A7C15 + i)
        files.append(ctx.encode_pcm(pcm, 44100, 128, bits_of("3#s%02d" % i))["mp3"])
    out = ctx.decode_streams(files, mlib.MP3S_PCM_I16)
    for i, (f, r) in enumerate(zip(files, out)):
        od = orc.decode(f)
        assert r["n_frames"] == od["n_frames"] and np.array_equal(r["bits"], od["bits"])
        assert zlib.crc32(r["pcm"].tobytes()) == zlib.crc32(orc.pcm_to_i16(od["pcm"]).tobytes()), i
    out64 = ctx.decode_streams(files[4:7], mlib.MP3S_PCM_F64)
    assert np.array_equal(out64[1]["pcm"], orc.decode(files[5])["pcm"])


# ------------------------------------------------------------------------------------------------ long messages
def test_long_messages_take_the_variant_path(ctx, mlib, orc):
    """messages above 1024 bits: the rate loop runs once per 3-bit pattern and the cursor walk picks each unit's pattern
    (mp3s_encode_pipeline.cpp encode_batch).  Lengths around the threshold, ending mid-stream (the units at the message end fall back
    to the exact re-run), ending on every offset inside a unit, and longer than the stream can hold."""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(700, seed=77)
    pcm[300 * 1152:330 * 1152] = 0                                   # silent units: no tables, stale addresses (E7)
    rng = np.random.default_rng(3)
    for nbits in (300, 1001, 1024, 1025, 1026, 1027, 3203, 6000, 24040):
        msg = rng.integers(0, 2, size=nbits).astype(np.uint8)
        r = ctx.encode_pcm(pcm, 44100, 128, msg)
        o = orc.encode(pcm, 44100, 128, msg)
        assert o["rc"] == 0 and r["mp3"] == o["mp3"], nbits
        assert r["hide_offset"] == o["hide_offset"] and r["too_long"] == bool(o["too_long"]), nbits
        d = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_I16)
        k = min(nbits, int(r["hide_offset"]))
        assert np.array_equal(d["bits"][:k], msg[:k]), nbits         # what went in comes out
    # other rates: the variant launches share the per-rate tables of the batch
    for rate, kbps in ((48000, 320), (32000, 64)):
        pcm2 = synth_pcm(300, rate=rate, seed=5)
        msg = rng.integers(0, 2, size=2500).astype(np.uint8)
        assert ctx.encode_pcm(pcm2, rate, kbps, msg)["mp3"] == orc.encode(pcm2, rate, kbps, msg)["mp3"], (rate, kbps)


@pytest.mark.gpu
def test_config4_one_million_frames_as_eight_blocks():
    """BASELINE configs[3] at size: 1 000 000 frames = 100 streams x 10 000 frames cut into 8 contiguous blocks of 125 000
    frames (the ranks of an 8-GPU node, played one after another on this device; block boundaries inside streams 12, 37, 62,
    87 pass the 136-byte carry).  Every stream reassembled from the blocks equals mp3s_hide_message on the whole stream, the
    oracle agrees byte for byte on a 1 % sample (tools/config4.py, a process of its own: it forks workers for the PCM)."""
    import json
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "config4.py")
    r = subprocess.run([sys.executable, tool, "100", "10000", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["frames"] == 1000000 and res["streams_equal_to_single_call"] == 100
    assert res["oracle_sample_streams_equal"] == res["oracle_sample_streams"] == 8 and res["oracle_sample_frames"] >= 10000
    assert res["pipe"]["slow"] == 0


@pytest.mark.gpu
def test_encode_transform_takes_pcm_from_any_int16_address(ctx, mlib):
    """k_enc_analysis stages its PCM tile 16 bytes per lane when the caller's device pointer allows it and sample by sample when it
    does not (`mp3s_encode_transform_dev` is C-ABI: any int16-aligned address): the same MDCT lines either way
    (encoder/MP3_Encoder.py:321-370, 652-758)."""
    import ctypes as C
    from synth_pcm import synth_pcm
    n = 40
    pcm = np.ascontiguousarray(synth_pcm(n, seed=23), dtype=np.int16)
    hdr = np.zeros(n, dtype=mlib.FRAME_HDR_DTYPE)
    hdr["nch"] = 2
    want = ctx.encode_transform(pcm, hdr)
    d_hdr = ctx.to_device(hdr)
    d_mdct = ctx.alloc(want.nbytes)
    d_pcm = ctx.alloc(pcm.nbytes + 64)
    try:
        for shift in (0, 4, 6, 18):                          # bytes: 16-byte aligned, and three addresses that are not
            at = C.c_void_p(d_pcm.value + shift)
            ctx.upload(at, pcm)
            mlib.check(mlib.lib().mp3s_encode_transform_dev(ctx.handle, at, d_hdr, n, d_mdct))
            ctx.sync()
            got = ctx.download(d_mdct, np.int32, want.shape)
            assert np.array_equal(got, want), shift
    finally:
        ctx.free(d_pcm); ctx.free(d_mdct); ctx.free(d_hdr)


@pytest.mark.gpu
def test_pure_tones_reach_the_longest_escapes(ctx, mlib, orc):
    """A full-scale sine in one channel and a full-scale square wave in the other put nearly all of a granule's bits into a few lines:
    quantised values of up to 8 000 (books 23 / 31, 13 linbits; the quantiser's float branch on every probe) -- the far end of the
    bit packer's escape fields and of the rate loop's tables (encoder/MP3_Encoder.py:389-415, 1452-1500).  Bytes, cursor, scfsi and
    the decode of the result against the oracle; with and without a message."""
    n = 24
    t = np.arange(n * 1152) / 44100.0
    left = np.rint(32767 * np.sin(2 * np.pi * 1000 * t)).astype(np.int16)
    right = np.where(np.sin(2 * np.pi * 220 * t) >= 0, 32767, -32768).astype(np.int16)
    pcm = np.ascontiguousarray(np.stack([left, right], axis=1))
    seen = 0
    for kbps, msg in ((320, None), (128, bits_of("6#tones.")), (64, None)):
        o = orc.encode(pcm, 44100, kbps, msg)
        r = ctx.encode_pcm(pcm, 44100, kbps, msg)
        assert o["rc"] == 0 and r["mp3"] == o["mp3"], kbps
        assert r["hide_offset"] == o["hide_offset"]
        assert np.array_equal(r["scfsi"], o["frames"]["scfsi"]), kbps
        d, od = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_F64), orc.decode(r["mp3"])   # ... and back: the decoder's escapes
        assert np.array_equal(d["pcm"], od["pcm"]) and np.array_equal(d["bits"], od["bits"]), kbps
        seen = max(seen, int(np.abs(np.asarray(o["ix"])).max()))
    assert seen > 4096                                                  # (the premise: 13-bit escapes were in the streams)
