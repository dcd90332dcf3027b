"""Host stages of the product (bit parsing / packing, no GPU) against the oracle and the goldens."""
import os

import numpy as np
import pytest


def test_parse_stream_testmp3(mlib, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_decode_testmp3.npz"))
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    p = mlib.parse_stream(data)
    assert p["n_frames"] == 36 and p["channels"] == 2 and p["sampling_rate"] == 44100 and p["bit_rate"] == 320000
    assert np.array_equal(p["is"], g["is"])
    assert np.array_equal(p["bits"], g["bits"])
    assert np.array_equal(p["table_select"], g["table_select"])
    assert np.array_equal(p["frame_size"], g["frame_size"])
    si = p["si"]
    assert np.array_equal(si["global_gain"], g["si_global_gain"])
    assert np.array_equal(si["scalefac_scale"], g["si_scale_fac_scale"])
    assert np.array_equal(si["block_type"], g["si_block_type"])
    assert np.array_equal(si["preflag"], g["si_pre_flag"])
    assert np.array_equal(si["scale_fac_l"], g["scale_fac_l"])
    assert np.array_equal(si["scale_fac_s"], g["scale_fac_s"])
    assert np.array_equal(si["sub_block_gain"], g["sub_block_gain"])
    assert p["dup_last_frame"] == 0 and not p["hdr"]["ms_stereo"].any()


def test_parse_stream_synth_and_edge_cases(mlib, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_synth128.npz"))
    data = g["mp3"].tobytes()
    p = mlib.parse_stream(data)
    o = orc.decode(data)
    assert np.array_equal(p["is"], o["is"]) and np.array_equal(p["bits"], o["bits"])
    assert np.array_equal(p["bits"], g["dec_bits"])
    # ID3v2 tag in front (ID3_Parser.py:95-131): 10-byte header + synch-safe size
    tag = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 20]) + b"\x00" * 20
    p2 = mlib.parse_stream(tag + data)
    assert p2["n_frames"] == p["n_frames"] and np.array_equal(p2["is"], p["is"])
    # garbage after the 10th frame: the reference appends the last PCM frame once more and stops (D12)
    cut = int(p["frame_size"][:10].sum())
    bad = data[:cut] + b"\x00" * 64
    p3 = mlib.parse_stream(bad)
    o3 = orc.decode(bad)
    assert p3["n_frames"] == 10 and p3["dup_last_frame"] == 1
    assert o3["rc"] == 0 and o3["n_frames"] == 10 and o3["pcm"].shape[0] == 11 * 1152
    # a truncated last frame is zero padded by get_bits (D12)
    p4 = mlib.parse_stream(data[:-100])
    o4 = orc.decode(data[:-100])
    assert p4["n_frames"] == o4["n_frames"] and np.array_equal(p4["is"], o4["is"])
    for tiny in (b"\x00", b"\xff\xfb", b"\xff\xfb\x90"):            # one byte; a sync with no header behind it
        with pytest.raises(mlib.Mp3sError):
            mlib.parse_stream(tiny)
        assert orc.decode(tiny)["rc"] != 0
    # no sync at the start: nothing is parsed (the reference then writes an empty WAV), not an error
    for junk in (b"\x00" * 3, b"\x12\x34", b"\x00" * 5000):
        pj = mlib.parse_stream(junk)
        oj = orc.decode(junk)
        assert oj["rc"] == 0 and pj["n_frames"] == oj["n_frames"] == 0 and len(pj["bits"]) == 0


def test_rate_frames_and_format_stream(mlib, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_synth128.npz"))
    o = orc.encode(g["pcm"], 44100, 128, g["hide_bits"])
    n = o["n_frames"]
    rf, pad = mlib.rate_frames(44100, 128, 2, n)
    assert np.array_equal(pad, g["padding"])
    assert set(rf["max_bits"].tolist()) <= {762, 764} and (rf["sr_idx"] == 0).all()
    gi = o["frames"]["gi"]
    gr = np.zeros(n * 4, dtype=mlib.GR_OUT_DTYPE)
    for f in range(n):
        for ch in range(2):
            for k in range(2):
                u = (f * 2 + ch) * 2 + k
                s = gi[f][k][ch]
                for a, b in (("part2_3_length", "part2_3_length"), ("big_values", "big_values"), ("count1", "count1"),
                             ("quantizer_step", "quantizerStepSize"), ("region0_count", "region0_count"),
                             ("region1_count", "region1_count"), ("count1table_select", "count1table_select")):
                    gr[u][a] = s[b]
                gr[u]["table_select"] = s["table_select"]
    # the oracle's part2_3_length already contains the stuffing; remove it from gr0/ch0 (plan a) to feed raw values
    raw = gr.copy()
    for f in range(n):
        mean_bits = (8 * (417 + int(pad[f])) - 288) // 2
        tot = sum(int(raw[(f * 2 + ch) * 2 + k]["part2_3_length"]) for ch in range(2) for k in range(2))
        assert tot == 2 * mean_bits            # every frame is filled completely (E6)
    mp3 = mlib.format_stream(44100, 128, o["ix"].astype(np.int16), gr, o["frames"]["scfsi"])
    assert mp3 == o["mp3"] == g["mp3"].tobytes()


def test_one_frame_host_decode_equals_the_stream_parser(mlib, golden_dir):
    """parse_scanned_frame (the host's answer for frames the device Huffman kernel flags) against the full stream parser,
    frame by frame, on every stream of the corpus the device would decode itself: same samples, same scalefactors the
    transforms read"""
    L = mlib.lib()
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    streams = [open(os.path.join(golden_dir, "test.mp3"), "rb").read(), np.load(os.path.join(golden_dir, "g6_synth128.npz"))["mp3"].tobytes()]
    streams += [g[n + "__mp3"].tobytes() for n in names]
    checked = 0
    for data in streams:
        s = mlib.scan_stream(data)
        if not s["gpu_ok"]:
            continue
        p = mlib.parse_stream(data)
        side, blob = np.ascontiguousarray(s["side"]), np.ascontiguousarray(s["blob"])
        nch = s["channels"]
        for f in range(s["n_frames"]):
            isv = np.zeros((2, 2, 576), dtype=np.int16)
            si = np.zeros((2, 2), dtype=mlib.GRANULE_SI_DTYPE)
            mlib.check(L.mp3s_debug_parse_scanned_frame(side[f:f + 1].ctypes.data, blob.ctypes.data, isv.ctypes.data, si.ctypes.data))
            assert np.array_equal(isv[:, :nch], p["is"][f][:, :nch]), f
            for gr in range(2):
                for ch in range(nch):
                    a, b = si[gr, ch], p["si"][f, gr, ch]
                    for k in ("global_gain", "scalefac_scale", "block_type", "mixed_block_flag", "preflag", "sub_block_gain"):
                        assert np.array_equal(a[k], b[k]), (f, gr, ch, k)
                    if b["block_type"] == 2:
                        assert np.array_equal(a["scale_fac_s"], b["scale_fac_s"]), (f, gr, ch)
                    else:
                        assert np.array_equal(a["scale_fac_l"], b["scale_fac_l"]), (f, gr, ch)
            checked += 1
    assert checked > 100
