"""Pins the ORACLE (oracle/*.c) against the goldens produced by the upstream reference
(tests/golden/gen_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_oracle_decode_testmp3(orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_decode_testmp3.npz"))
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    r = orc.decode(data)
    assert r["rc"] == 0 and r["n_frames"] == 36 and r["channels"] == 2
    assert r["sampling_rate"] == 44100 and r["bit_rate"] // 1000 == int(g["kbps"]) == 320
    assert np.array_equal(r["is"], g["is"])
    assert np.array_equal(r["bits"], g["bits"]) and len(r["bits"]) == 419
    assert np.array_equal(r["pcm"][:4 * 1152], g["pcm_head"])          # float64, bit for bit
    assert sha(np.ascontiguousarray(r["pcm"]).tobytes()) == bytes(g["pcm_sha256"]).decode()
    i16 = orc.pcm_to_i16(r["pcm"])
    assert sha(i16.tobytes()) == bytes(g["pcm_i16_sha256"]).decode()
    wav = orc.wav_bytes(i16, 44100)
    assert len(wav) == int(g["wav_len"]) and sha(wav) == bytes(g["wav_sha256"]).decode()
    fr = r["frames"]
    for k in ["part2_3_length", "big_value", "global_gain", "scale_fac_compress", "window_switching", "block_type",
              "mixed_block_flag", "region0_count", "region1_count", "pre_flag", "scale_fac_scale", "count1table_select"]:
        assert np.array_equal(fr[k], g["si_" + k]), k
    for k in ["table_select", "sub_block_gain", "scale_fac_l", "scale_fac_s", "scfsi", "main_data_begin", "frame_size"]:
        assert np.array_equal(fr[k], g[k]), k


@pytest.mark.parametrize("name", ["plain320", "hide_ddd320"])
def test_oracle_encode_testmp3(orc, golden_dir, name):
    pcm = np.load(os.path.join(golden_dir, "g3_testmp3_wav_pcm.npz"))["pcm"]
    g = np.load(os.path.join(golden_dir, f"g3_encode_{name}.npz"))
    hide = g["hide_bits"] if "hide_bits" in g.files else None
    r = orc.encode(pcm, 44100, 320, hide)
    assert r["rc"] == 0 and r["n_frames"] == 36
    assert len(r["mp3"]) == int(g["mp3_len"]) and sha(r["mp3"]) == bytes(g["mp3_sha256"]).decode()
    assert np.array_equal(r["mdct_freq"][:4], g["mdct_freq"])
    assert np.array_equal(r["ix"][:4], g["ix"])
    fields = bytes(g["gi_fields"]).decode().split(",")
    for i, k in enumerate(fields):
        assert np.array_equal(r["frames"]["gi"][k], g["gi"][..., i]), k
    assert np.array_equal(r["frames"]["gi"]["table_select"], g["table_select"])
    assert np.array_equal(r["frames"]["scfsi"], g["scfsi"])
    assert np.array_equal(r["frames"]["written"], g["written"])
    assert np.array_equal(r["frames"]["hide_off"], g["hide_off"])
    assert np.array_equal(r["frames"]["padding"], g["padding"])
    assert bool(r["too_long"]) == bool(int(g["too_long"]))


def test_oracle_facade_hashes(orc, golden_dir):
    """hide -> reveal -> clear -> reveal through oracle decode/encode equals the reference facade run."""
    fac = json.load(open(os.path.join(golden_dir, "g3_facade.json")))
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()

    def bits_of(s):
        return np.frombuffer("".join(format(b, "08b") for b in s.encode()).encode(), dtype=np.uint8) - ord("0")

    d = orc.decode(data)
    pcm = orc.pcm_to_i16(d["pcm"])
    kbps = d["bit_rate"] // 1000
    h = orc.encode(pcm, d["sampling_rate"], kbps, bits_of("3#ddd"))
    assert sha(h["mp3"]) == fac["hide_sha256"] and bool(h["too_long"]) == fac["too_long"]
    d2 = orc.decode(h["mp3"])
    c = orc.encode(orc.pcm_to_i16(d2["pcm"]), d2["sampling_rate"], d2["bit_rate"] // 1000, None)
    assert sha(c["mp3"]) == fac["cleared_sha256"]
    long = orc.encode(pcm, d["sampling_rate"], kbps, bits_of("300#" + "ddd" * 100))
    assert sha(long["mp3"]) == fac["hide_long_sha256"] and bool(long["too_long"]) == fac["too_long_300"]


def _chain_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_decode_chain.npz"))
    return g, sorted({k.split("__")[0] for k in g.files})


def test_oracle_decode_chain(orc, golden_dir):
    g, names = _chain_cases(golden_dir)
    L = orc.lib()
    for name in names:
        hdr, isv, pcm_ref = g[name + "__hdr"], g[name + "__is"], g[name + "__pcm"]
        sr_idx = int((hdr[2] >> 2) & 3)
        mode = int(hdr[3] >> 6)
        ms = mode == 1 and (hdr[3] & 0x20)
        nch = 1 if mode == 3 else 2
        prev = np.zeros((2, 32 * 18))
        fifo = np.zeros((2, 1024))
        for f in range(isv.shape[0]):
            smp = isv[f].astype(np.float64)
            for gr in range(2):
                for ch in range(nch):
                    s = np.ascontiguousarray(smp[gr, ch])
                    L.orc_requantize(s, int(g[name + "__gg"][f][gr][ch]), int(g[name + "__sfs"][f][gr][ch]),
                                     int(g[name + "__bt"][f][gr][ch]), int(g[name + "__mixed"][f][gr][ch]),
                                     int(g[name + "__pre"][f][gr][ch]),
                                     np.ascontiguousarray(g[name + "__sbg"][f][gr][ch], dtype=np.int32),
                                     np.ascontiguousarray(g[name + "__sfl"][f][gr][ch], dtype=np.int32),
                                     np.ascontiguousarray(g[name + "__sfsh"][f][gr][ch], dtype=np.int32), sr_idx)
                    smp[gr, ch] = s
                if ms:
                    a, b = np.ascontiguousarray(smp[gr, 0]), np.ascontiguousarray(smp[gr, 1])
                    L.orc_ms_stereo(a, b)
                    smp[gr, 0], smp[gr, 1] = a, b
                for ch in range(nch):
                    s = np.ascontiguousarray(smp[gr, ch])
                    bt, mx = int(g[name + "__bt"][f][gr][ch]), int(g[name + "__mixed"][f][gr][ch])
                    if bt == 2 or mx:
                        L.orc_reorder(s, sr_idx)
                    else:
                        L.orc_alias_reduction(s)
                    L.orc_imdct(s, bt, prev[ch])
                    L.orc_frequency_inversion(s)
                    L.orc_synth_filter_bank(s, fifo[ch])
                    smp[gr, ch] = s
            out = np.zeros((1152, nch))
            for gr in range(2):
                for ch in range(nch):
                    out[gr * 576:(gr + 1) * 576, ch] = smp[gr, ch]
            assert np.array_equal(out, pcm_ref[f]), (name, f)


def test_oracle_encode_stages(orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_encode_stages.npz"))
    L = orc.lib()
    x = np.zeros(512, dtype=np.int32)
    off = np.zeros(1, dtype=np.int32)
    for k in range(6):
        for i in range(31, -1, -1):
            x[i + off[0]] = np.int32(g["wf_pcm"][k][31 - i]) << 16
        s = np.zeros(32, dtype=np.int32)
        L.orc_enc_window_filter_subband(s, x, off)
        assert np.array_equal(s, g["wf_sb"][k])
    xr = g["q_xr"]
    xrabs = np.abs(xr.astype(np.int64)).astype(np.int32)
    for n, st in enumerate(g["q_steps"]):
        ix = np.zeros(576, dtype=np.int32)
        m = L.orc_enc_quantize(ix, int(st), int(xrabs.max()), xr, xrabs)
        assert m == g["q_max"][n]
        if m != 16384:
            assert np.array_equal(ix, g["q_ix"][n])


def test_oracle_synth128(orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_synth128.npz"))
    r = orc.encode(g["pcm"], 44100, 128, g["hide_bits"])
    assert r["rc"] == 0 and r["mp3"] == g["mp3"].tobytes()
    fields = bytes(g["gi_fields"]).decode().split(",")
    for i, k in enumerate(fields):
        assert np.array_equal(r["frames"]["gi"][k], g["gi"][..., i]), k
    assert np.array_equal(r["frames"]["gi"]["table_select"], g["table_select"])
    assert np.array_equal(r["frames"]["hide_off"], g["hide_off"])
    assert np.array_equal(r["ix"], g["ix"])
    d = orc.decode(g["mp3"].tobytes())
    assert d["rc"] == 0 and np.array_equal(d["bits"], g["dec_bits"])
    assert np.array_equal(d["bits"][:len(g["hide_bits"])], g["hide_bits"])
    assert sha(np.ascontiguousarray(d["pcm"]).tobytes()) == bytes(g["dec_pcm_sha256"]).decode()


def test_oracle_reference_errors(orc):
    """inputs the reference itself cannot process are reported, not mis-encoded"""
    pcm = np.zeros((1152, 1), dtype=np.int16)
    assert orc.encode(pcm, 44100, 128)["rc"] == -2          # mono
    pcm = np.zeros((1152 + 100, 2), dtype=np.int16)
    assert orc.encode(pcm, 44100, 128)["rc"] == -2          # partial last frame
