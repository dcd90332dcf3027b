"""Mutated streams: bit flips in headers, side info and main data of tests/test.mp3 and of a synthesised corpus stream.
The reference keeps going through most damage (false syncs are parsed as whatever they claim to be, a bad header ends
the stream with the last frame repeated, only missing band tables and array overruns raise), so for every mutant the
library must reach the oracle's verdict -- the same frames, spectra and stego bits, or an error where the oracle (=
the reference's IndexError / ZeroDivisionError) has one -- and must never crash."""
import os

import numpy as np
import pytest


def mutants(data, n, seed, span=None):
    rng = np.random.default_rng(seed)
    lo, hi = span if span else (0, len(data))
    for _ in range(n):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            b[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
        yield bytes(b)


def frame_starts(mlib, data):
    fs = mlib.parse_stream(data)["frame_size"]
    first = len(data) - int(fs.sum())            # test.mp3 / the corpus streams end with their last frame
    return first + np.concatenate([[0], np.cumsum(fs)[:-1]])


def header_mutants(mlib, data, n, seed):
    """flips confined to the 4 header + 32 side-info bytes of random frames (the fields that steer the decoder)"""
    rng = np.random.default_rng(seed)
    starts = frame_starts(mlib, data)
    for _ in range(n):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(starts[int(rng.integers(0, len(starts)))]) + int(rng.integers(0, 36))] ^= 1 << int(rng.integers(0, 8))
        yield bytes(b)


def check_host(mlib, orc, m):
    o = orc.decode(m)
    try:
        p = mlib.parse_stream(m)
    except mlib.Mp3sError as e:
        assert e.code in (mlib.E_MALFORMED, mlib.E_UNSUPPORTED)
        assert o["rc"] != 0                      # (the byte-level scan alone may accept it: overruns inside the Huffman
        return None                              #  data are found by the device kernel, test below)
    assert o["rc"] == 0
    assert p["n_frames"] == o["n_frames"] and np.array_equal(p["is"], o["is"]) and np.array_equal(p["bits"], o["bits"])
    return o


def check_frame_by_frame(mlib, m, o):
    """the one-frame host decode (what the device pipeline falls back to for frames its Huffman kernel flags) on every
    frame of a stream the device would decode itself: the oracle's samples"""
    s = mlib.scan_stream(m)
    if not s["gpu_ok"] or s["n_frames"] == 0:
        return 0
    side, blob, L = np.ascontiguousarray(s["side"]), np.ascontiguousarray(s["blob"]), mlib.lib()
    isv = np.zeros((2, 2, 576), dtype=np.int16)
    si = np.zeros((2, 2), dtype=mlib.GRANULE_SI_DTYPE)
    nch = s["channels"]
    for f in range(s["n_frames"]):
        mlib.check(L.mp3s_debug_parse_scanned_frame(side[f:f + 1].ctypes.data, blob.ctypes.data, isv.ctypes.data, si.ctypes.data))
        assert np.array_equal(isv[:, :nch], o["is"][f][:, :nch]), f
    return s["n_frames"]


def test_host_parser_reaches_the_oracles_verdict(mlib, orc, golden_dir):
    with open(os.path.join(golden_dir, "test.mp3"), "rb") as f:
        data = f.read()
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    ok = bad = 0
    cases = list(header_mutants(mlib, data, 250, 1)) + list(mutants(data, 100, 2))
    for n in names[:3]:
        cases += list(header_mutants(mlib, g[n + "__mp3"].tobytes(), 40, 3)) + list(mutants(g[n + "__mp3"].tobytes(), 20, 4))
    frames = 0
    for m in cases:
        o = check_host(mlib, orc, m)
        ok += o is not None
        bad += o is None
        if o is not None:
            frames += check_frame_by_frame(mlib, m, o)
    assert ok > 300 and bad > 5                   # both outcomes are exercised
    assert frames > 5000                          # ... and the one-frame host decode on most frames of the accepted ones


@pytest.mark.gpu
def test_device_decode_of_mutants_matches_the_oracle(ctx, mlib, orc, golden_dir):
    with open(os.path.join(golden_dir, "test.mp3"), "rb") as f:
        data = f.read()
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    cases = list(header_mutants(mlib, data, 120, 11)) + list(mutants(data, 40, 12))
    for n in names[:3]:
        cases += list(header_mutants(mlib, g[n + "__mp3"].tobytes(), 25, 13))
    ok = 0
    for m in cases:
        o = orc.decode(m)
        try:
            r = ctx.decode_stream(m, mlib.MP3S_PCM_F64)
        except mlib.Mp3sError as e:
            assert e.code in (mlib.E_MALFORMED, mlib.E_UNSUPPORTED) and o["rc"] != 0
            continue
        assert o["rc"] == 0
        assert r["n_frames"] == o["n_frames"] and np.array_equal(r["bits"], o["bits"])
        assert r["pcm"].shape == o["pcm"].shape
        assert np.array_equal(r["pcm"].view(np.uint64), np.ascontiguousarray(o["pcm"]).view(np.uint64))   # NaNs compare too
        ok += 1
    assert ok > 100


@pytest.mark.gpu
def test_encoder_on_extreme_signals(ctx, mlib, orc):
    """full-scale noise, square waves, impulses, DC, near silence -- at the tightest and the widest bit budget: the rate
    loop's float path (ln >= 10000), the early out, empty granules and the quantizer-step guard all against the oracle"""
    rng = np.random.default_rng(5)
    n = 24 * 1152
    t = np.arange(n)
    sig = {
        "noise": rng.integers(-32768, 32768, size=(n, 2)),
        "square": np.stack([np.where((t // 7) % 2, 32767, -32768), np.where((t // 50) % 2, -32768, 32767)], axis=1),
        "impulse": np.zeros((n, 2)),
        "dc": np.full((n, 2), 32767),
        "quiet": rng.integers(-1, 2, size=(n, 2)),
        "burst": np.concatenate([np.zeros((n // 2, 2)), rng.integers(-32768, 32768, size=(n - n // 2, 2))]),
        "left_only": np.stack([rng.integers(-20000, 20000, size=n), np.zeros(n)], axis=1),
    }
    sig["impulse"][5000] = (32767, -32768)
    msg = np.frombuffer(b"1" * 30 + b"0" * 30, dtype=np.uint8) - ord("0")
    for name, pcm in sig.items():
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        for rate, kbps in ((44100, 32), (48000, 320), (32000, 128)):
            o = orc.encode(pcm, rate, kbps, msg)
            try:
                r = ctx.encode_pcm(pcm, rate, kbps, msg)
            except mlib.Mp3sError as e:
                assert e.code == mlib.E_STEP_RANGE and o["rc"] != 0, (name, rate, kbps)
                continue
            assert o["rc"] == 0, (name, rate, kbps)
            assert r["mp3"] == o["mp3"], (name, rate, kbps)
            assert r["hide_offset"] == o["hide_offset"]
            d = ctx.decode_stream(r["mp3"], mlib.MP3S_PCM_F64)
            assert np.array_equal(d["pcm"], orc.decode(r["mp3"])["pcm"]), (name, rate, kbps)
