#!/usr/bin/env python3
"""Golden-vector generator: runs the upstream Python reference (build container
only, via refshim.py) and writes small fixtures next to this file.

    python tests/golden/gen_golden.py [tables] [decode] [encode] [stages] [synth]

Fixtures are DATA (inputs + the reference's outputs); no reference source is
copied.  Everything the `-m gpu` tests / bench / smoke need on the GPU box is
in the .npz/.bin files written here -- the reference itself never travels.

Sections
  tables  G1  every constant table the hot path uses, as evaluated by the
              reference (decoder/tables.py, encoder/tables.py, the njit init
              functions) -- used to cross-check our regenerated tables.
  decode  G2  tests/test.mp3 through Decoder: per-frame Huffman-decoded `is`,
              side info, stego bits, float64 PCM, WAV bytes hash.
  encode  G3  that WAV through Encoder @320 (no hide / hide 'ddd' / cleared):
              MP3 hashes, per g*c GrInfo, scfsi, mdct_freq, ix.
  stages  G4/G5 seeded random inputs through the individual stage functions.
  synth   G6  64 synthetic 44.1 kHz stereo frames encoded @128 kbps with a
              40-bit message, and decoded back.
"""
import hashlib
import os
import struct
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from refshim import load_reference  # noqa: E402

ref = load_reference()
from mp3stego.decoder import Frame as RF  # noqa: E402
from mp3stego.decoder import tables as RDT  # noqa: E402
from mp3stego.decoder import util as RDU  # noqa: E402
from mp3stego.decoder.decoder import Decoder as RDecoder  # noqa: E402
from mp3stego.encoder import MP3_Encoder as RE  # noqa: E402
from mp3stego.encoder import tables as RET  # noqa: E402
from mp3stego.encoder import util as REU  # noqa: E402
from mp3stego.encoder.encoder import Encoder as REncoder  # noqa: E402
from mp3stego.steganography import Steganography as RSteg, str_to_binary_str  # noqa: E402

WORK = "/tmp/mp3s_golden_work"
os.makedirs(WORK, exist_ok=True)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# --------------------------------------------------------------------------
# G1 tables
# --------------------------------------------------------------------------
def gen_tables():
    out = {}
    out["synth_window"] = np.asarray(RDT.synth_window, dtype=np.float64)
    out["pre_tab"] = np.asarray(RDT.pre_tab, dtype=np.int32)
    out["slen"] = np.asarray(RDT.slen, dtype=np.int32)
    for sr in ("32", "44", "48"):
        out[f"bw_long_{sr}"] = np.asarray(getattr(RDT.band_width_table, f"long_{sr}"), dtype=np.int32)
        out[f"bw_short_{sr}"] = np.asarray(getattr(RDT.band_width_table, f"short_{sr}"), dtype=np.int32)
        out[f"bi_long_{sr}"] = np.asarray(getattr(RDT.band_index_table, f"long_{sr}"), dtype=np.int32)
        out[f"bi_short_{sr}"] = np.asarray(getattr(RDT.band_index_table, f"short_{sr}"), dtype=np.int32)
    out["quad_value"] = np.asarray(RDT.quad_table_1.value, dtype=np.int32)
    out["quad_hlen"] = np.asarray(RDT.quad_table_1.h_len, dtype=np.int32)
    out["quad_hcod"] = np.asarray(RDT.quad_table_1.h_cod, dtype=np.uint32)
    out["big_value_linbit"] = np.asarray(RDT.big_value_linbit, dtype=np.int32)
    out["big_value_max"] = np.asarray(RDT.big_value_max, dtype=np.int32)
    for n in range(32):
        out[f"dec_hft_{n}"] = np.asarray(RDT.big_value_table[n], dtype=np.uint32)
    out["sine_block"] = RF.create_sine_block()
    out["synth_matrix"] = RF.init_synth_filter_bank_block()
    out["H0"] = np.asarray(sorted(RDU.H0), dtype=np.int32)
    # encoder
    out["enwindow"] = np.asarray(RET.enwindow, dtype=np.int32)
    out["slen1_tab"] = np.asarray(RET.slen1_tab, dtype=np.int32)
    out["slen2_tab"] = np.asarray(RET.slen2_tab, dtype=np.int32)
    out["enc_sfb_index"] = np.asarray(REU.scale_fact_band_index, dtype=np.int32)
    out["subdv_table"] = np.asarray(RET.subdv_table, dtype=np.int32)
    out["mdct_cs"] = np.asarray([getattr(RET, f"MDCT_CS{i}") for i in range(8)], dtype=np.int32)
    out["mdct_ca"] = np.asarray([getattr(RET, f"MDCT_CA{i}") for i in range(8)], dtype=np.int32)
    meta = []
    for n, t in enumerate(RET.huffman_table):
        meta.append([t.x_len, t.y_len, t.lin_bits, t.lin_max])
        if t.table is not None:
            out[f"enc_hcod_{n}"] = np.asarray(t.table, dtype=np.uint32)
            out[f"enc_hlen_{n}"] = np.asarray(t.h_len, dtype=np.int32)
    out["enc_huff_meta"] = np.asarray(meta, dtype=np.int32)
    tr = np.zeros((32, 2), dtype=np.int32)
    for (t, b), v in RE.IDX_TO_TRANSFORM_HUF.items():
        tr[t, b] = v
    out["idx_to_transform_huf"] = tr
    out["bit_rates"] = np.asarray(REU.BIT_RATES, dtype=np.int32)
    out["sample_rates"] = np.asarray(REU.SAMPLE_RATES, dtype=np.int32)
    # the encoder's init-time tables need a live MP3Encoder
    wav = os.path.join(WORK, "tiny.wav")
    write_wav(wav, np.zeros((1152, 2), dtype=np.int16), 44100)
    enc = RE.MP3Encoder(RE.WavReader(wav, 128))
    out["enc_fl"] = enc._MP3Encoder__sub_band.fl.copy()
    out["enc_cos_l"] = enc._MP3Encoder__mdct.cos_l.copy()
    out["enc_steptab"] = enc._MP3Encoder__l3loop.steptab.copy()
    out["enc_steptabi"] = enc._MP3Encoder__l3loop.steptabi.copy()
    out["enc_int2idx"] = enc._MP3Encoder__l3loop.int2idx.copy()
    np.savez_compressed(os.path.join(HERE, "g1_tables.npz"), **out)
    print("tables:", len(out), "arrays")


def write_wav(path, data, rate):
    """44-byte canonical PCM WAV (what scipy.io.wavfile.write emits for int16)."""
    data = np.ascontiguousarray(data, dtype="<i2")
    nch = 1 if data.ndim == 1 else data.shape[1]
    nbytes = data.nbytes
    hdr = b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, nch, rate, rate * nch * 2, nch * 2, 16) + b"data" + struct.pack("<I", nbytes)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(data.tobytes())


# --------------------------------------------------------------------------
# instrumented decode
# --------------------------------------------------------------------------
SI_FIELDS = ["part2_3_length", "big_value", "global_gain", "scale_fac_compress", "window_switching",
             "block_type", "mixed_block_flag", "region0_count", "region1_count", "pre_flag",
             "scale_fac_scale", "count1table_select"]


def decode_instrumented(mp3_path, wav_path, keep_pcm_frames=None):
    """Run the reference Decoder on mp3_path, recording per-frame artefacts."""
    rec = {"is": [], "pcm": [], "si": {k: [] for k in SI_FIELDS}, "table_select": [], "sub_block_gain": [],
           "scale_fac_l": [], "scale_fac_s": [], "scfsi": [], "main_data_begin": [], "hdr": [], "frame_size": []}
    cur_is = {}
    orig_req = RF.re_quantize
    orig_init = RF.Frame.init_frame_params

    def req_wrap(gr, ch, *a):
        samples = a[9]
        cur_is[(gr, ch)] = samples[gr][ch].copy()
        return orig_req(gr, ch, *a)

    def init_wrap(self, buffer, file_data, curr_offset):
        cur_is.clear()
        rec["hdr"].append(list(buffer[:4]))
        orig_init(self, buffer, file_data, curr_offset)
        nch = self._Frame__header.channels
        isf = np.zeros((2, 2, 576), dtype=np.int16)
        for (gr, ch), v in cur_is.items():
            isf[gr, ch] = v.astype(np.int16)
        rec["is"].append(isf)
        rec["pcm"].append(self.pcm.copy())
        si = self.side_info
        for k in SI_FIELDS:
            rec["si"][k].append(np.asarray(getattr(si, k)).astype(np.int32).copy())
        rec["table_select"].append(np.asarray(si.table_select).astype(np.int32).copy())
        rec["sub_block_gain"].append(np.asarray(si.sub_block_gain).astype(np.int32).copy())
        rec["scale_fac_l"].append(np.asarray(si.scale_fac_l).astype(np.int32).copy())
        rec["scale_fac_s"].append(np.asarray(si.scale_fac_s).astype(np.int32).copy())
        rec["scfsi"].append(np.asarray(si.scfsi).astype(np.int32).copy())
        rec["main_data_begin"].append(int(si.main_data_begin))
        rec["frame_size"].append(int(self.frame_size))

    RF.re_quantize = req_wrap
    RF.Frame.init_frame_params = init_wrap
    try:
        dec = RDecoder(mp3_path, wav_path)
        kbps = dec.decode(quiet=True)
        bits = dec._Decoder__parser.output_bits
    finally:
        RF.re_quantize = orig_req
        RF.Frame.init_frame_params = orig_init
    out = {
        "kbps": np.int32(kbps),
        "bits": np.frombuffer(bits.encode(), dtype=np.uint8) - ord("0"),
        "is": np.stack(rec["is"]),
        "hdr": np.asarray(rec["hdr"], dtype=np.uint8),
        "frame_size": np.asarray(rec["frame_size"], dtype=np.int32),
        "main_data_begin": np.asarray(rec["main_data_begin"], dtype=np.int32),
        "table_select": np.stack(rec["table_select"]),
        "sub_block_gain": np.stack(rec["sub_block_gain"]),
        "scale_fac_l": np.stack(rec["scale_fac_l"]),
        "scale_fac_s": np.stack(rec["scale_fac_s"]),
        "scfsi": np.stack(rec["scfsi"]),
    }
    for k in SI_FIELDS:
        out["si_" + k] = np.stack(rec["si"][k])
    pcm = np.concatenate(rec["pcm"], axis=0)
    out["pcm_sha256"] = np.frombuffer(sha(np.ascontiguousarray(pcm).tobytes()).encode(), dtype=np.uint8)
    n = len(rec["pcm"]) if keep_pcm_frames is None else keep_pcm_frames
    out["pcm_head"] = np.concatenate(rec["pcm"][:n], axis=0)
    out["pcm_i16_sha256"] = np.frombuffer(
        sha((pcm * 32767).astype(np.int16).tobytes()).encode(), dtype=np.uint8)
    with open(wav_path, "rb") as f:
        wav = f.read()
    out["wav_sha256"] = np.frombuffer(sha(wav).encode(), dtype=np.uint8)
    out["wav_len"] = np.int64(len(wav))
    return out, pcm


def gen_decode():
    t0 = time.time()
    mp3 = os.path.join(HERE, "test.mp3")
    wav = os.path.join(WORK, "test.wav")
    out, pcm = decode_instrumented(mp3, wav, keep_pcm_frames=4)
    np.savez_compressed(os.path.join(HERE, "g2_decode_testmp3.npz"), **out)
    print("decode: frames", out["is"].shape[0], "bits", len(out["bits"]), "kbps", int(out["kbps"]),
          "wav sha", bytes(out["wav_sha256"]).decode()[:16], f"{time.time() - t0:.1f}s")


# --------------------------------------------------------------------------
# instrumented encode
# --------------------------------------------------------------------------
GI_FIELDS = ["part2_3_length", "big_values", "count1", "global_gain", "scale_fac_compress", "region0_count",
             "region1_count", "preflag", "scale_fac_scale", "count1table_select", "part2_length", "address1",
             "address2", "address3", "quantizerStepSize"]


def encode_instrumented(wav_path, mp3_path, bitrate, hide_str="", keep_frames=None):
    rec = {"gi": [], "ts": [], "scfsi": [], "mdct": [], "ix": [], "written": [], "hoff": [], "padding": []}
    orig = RE.MP3Encoder._MP3Encoder__encode_buffer_internal

    def wrap(self):
        written, data = orig(self)
        nfr = len(rec["written"])
        gi = np.zeros((2, 2, len(GI_FIELDS)), dtype=np.int32)
        ts = np.zeros((2, 2, 3), dtype=np.int32)
        for gr in range(2):
            for ch in range(2):
                tt = self._MP3Encoder__side_info.gr[gr].ch[ch].tt
                gi[gr, ch] = [int(getattr(tt, k)) for k in GI_FIELDS]
                ts[gr, ch] = tt.table_select
        rec["gi"].append(gi)
        rec["ts"].append(ts)
        rec["scfsi"].append(self._MP3Encoder__side_info.scfsi.copy())
        if keep_frames is None or nfr < keep_frames:
            rec["mdct"].append(self._MP3Encoder__mdct_freq.copy())
            rec["ix"].append(self._MP3Encoder__l3_enc.copy())
        rec["written"].append(written)
        rec["hoff"].append(self.hide_str_offset)
        rec["padding"].append(self._MP3Encoder__mpeg.padding)
        return written, data

    RE.MP3Encoder._MP3Encoder__encode_buffer_internal = wrap
    try:
        enc = REncoder(wav_path, mp3_path, bitrate=bitrate, hide_str=hide_str)
        too_long = enc.encode(quiet=True)
    finally:
        RE.MP3Encoder._MP3Encoder__encode_buffer_internal = orig
    with open(mp3_path, "rb") as f:
        mp3 = f.read()
    out = {
        "too_long": np.int32(bool(too_long)),
        "gi": np.stack(rec["gi"]), "table_select": np.stack(rec["ts"]), "scfsi": np.stack(rec["scfsi"]),
        "mdct_freq": np.stack(rec["mdct"]), "ix": np.stack(rec["ix"]).astype(np.int16),
        "written": np.asarray(rec["written"], dtype=np.int32),
        "hide_off": np.asarray(rec["hoff"], dtype=np.int32),
        "padding": np.asarray(rec["padding"], dtype=np.int32),
        "mp3_sha256": np.frombuffer(sha(mp3).encode(), dtype=np.uint8),
        "mp3_len": np.int64(len(mp3)),
        "gi_fields": np.frombuffer(",".join(GI_FIELDS).encode(), dtype=np.uint8),
    }
    return out, mp3


def gen_encode():
    wav = os.path.join(WORK, "test.wav")
    if not os.path.exists(wav):
        decode_instrumented(os.path.join(HERE, "test.mp3"), wav)
    with open(wav, "rb") as f:
        wav_bytes = f.read()
    # the int16 PCM of the reference decode is a fixture in its own right
    np.savez_compressed(os.path.join(HERE, "g3_testmp3_wav_pcm.npz"),
                        pcm=np.frombuffer(wav_bytes[44:], dtype="<i2").reshape(-1, 2), rate=np.int32(44100))
    t0 = time.time()
    out, mp3_plain = encode_instrumented(wav, os.path.join(WORK, "plain.mp3"), 320, "", keep_frames=4)
    np.savez_compressed(os.path.join(HERE, "g3_encode_plain320.npz"), **out)
    print("encode plain:", bytes(out["mp3_sha256"]).decode()[:16], int(out["mp3_len"]), f"{time.time() - t0:.1f}s")
    t0 = time.time()
    hide = str_to_binary_str("3#ddd")
    out, mp3_hide = encode_instrumented(wav, os.path.join(WORK, "hide.mp3"), 320, hide, keep_frames=4)
    out["hide_bits"] = np.frombuffer(hide.encode(), dtype=np.uint8) - ord("0")
    np.savez_compressed(os.path.join(HERE, "g3_encode_hide_ddd320.npz"), **out)
    with open(os.path.join(HERE, "g3_hide_ddd.mp3"), "wb") as f:
        f.write(mp3_hide)
    print("encode hide:", bytes(out["mp3_sha256"]).decode()[:16], f"{time.time() - t0:.1f}s")
    # the full facade round trip: hide -> reveal -> clear -> reveal (reference tests 2-5)
    t0 = time.time()
    st = RSteg(quiet=True)
    src = os.path.join(WORK, "t.mp3")
    with open(src, "wb") as f, open(os.path.join(HERE, "test.mp3"), "rb") as g:
        f.write(g.read())
    too_long = st.hide_message(src, os.path.join(WORK, "h.mp3"), "ddd")
    st.reveal_massage(os.path.join(WORK, "h.mp3"), os.path.join(WORK, "r.txt"))
    st.clear_file(os.path.join(WORK, "h.mp3"), os.path.join(WORK, "c.mp3"))
    st.reveal_massage(os.path.join(WORK, "c.mp3"), os.path.join(WORK, "rc.txt"))
    too_long2 = st.hide_message(src, os.path.join(WORK, "h2.mp3"), "ddd" * 100)
    st.reveal_massage(os.path.join(WORK, "h2.mp3"), os.path.join(WORK, "r2.txt"))
    res = {
        "hide_sha256": sha(open(os.path.join(WORK, "h.mp3"), "rb").read()),
        "cleared_sha256": sha(open(os.path.join(WORK, "c.mp3"), "rb").read()),
        "hide_long_sha256": sha(open(os.path.join(WORK, "h2.mp3"), "rb").read()),
        "revealed": open(os.path.join(WORK, "r.txt"), "rb").read().decode("utf-8"),
        "revealed_cleared": open(os.path.join(WORK, "rc.txt"), "rb").read().decode("utf-8"),
        "revealed_long": open(os.path.join(WORK, "r2.txt"), "rb").read().decode("utf-8"),
        "too_long": bool(too_long), "too_long_300": bool(too_long2),
    }
    import json
    with open(os.path.join(HERE, "g3_facade.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("facade:", res, f"{time.time() - t0:.1f}s")


# --------------------------------------------------------------------------
# G4  decode transform chain on seeded random spectra (short / mixed / MS / 3 rates)
# --------------------------------------------------------------------------
def header_bytes(sr_code, mode, mode_ext, bitrate_idx=9, pad=0):
    return [0xFF, 0xFB, (bitrate_idx << 4) | (sr_code << 2) | (pad << 1), (mode << 6) | (mode_ext << 4)]


def run_chain(rng, n_frames, sr_code, mode, mode_ext, bt_choices, allow_mixed, max_abs):
    """Drive the reference's own transform functions the way Frame.init_frame_params
    does (decoder/Frame.py:265-286) on synthetic Huffman-decoded input."""
    fr = RF.Frame()
    fr.init_header_params(header_bytes(sr_code, mode, mode_ext))
    hdr = fr._Frame__header
    si = fr.side_info
    nch = hdr.channels
    rec = {k: [] for k in ("is", "gg", "sfs", "bt", "mixed", "pre", "sbg", "sfl", "sfsh", "pcm")}
    for f in range(n_frames):
        isv = np.zeros((2, 2, 576), dtype=np.int16)
        for gr in range(2):
            for ch in range(nch):
                n_big = int(rng.integers(0, 577))
                v = np.zeros(576, dtype=np.int64)
                mag = rng.integers(0, max_abs + 1, size=n_big)
                small = rng.random(n_big) < 0.8
                mag = np.where(small, rng.integers(0, 16, size=n_big), mag)
                v[:n_big] = mag * rng.choice([-1, 1], size=n_big)
                isv[gr, ch] = v
                bt = int(rng.choice(bt_choices))
                si.block_type[gr][ch] = bt
                si.window_switching[gr][ch] = bt != 0
                si.mixed_block_flag[gr][ch] = bool(allow_mixed and bt != 0 and rng.random() < 0.5)
                si.global_gain[gr][ch] = int(rng.integers(90, 200))
                si.scale_fac_scale[gr][ch] = int(rng.integers(0, 2))
                si.pre_flag[gr][ch] = int(rng.integers(0, 2))
                si.sub_block_gain[gr][ch][:] = rng.integers(0, 8, size=3)
                si.scale_fac_l[gr][ch][:21] = rng.integers(0, 16, size=21)
                si.scale_fac_l[gr][ch][21] = 0
                si.scale_fac_s[gr][ch][:, :12] = rng.integers(0, 16, size=(3, 12))
                si.scale_fac_s[gr][ch][:, 12] = 0
        fr._Frame__samples[:] = isv.astype(np.float64)
        fr._Frame__pcm = np.zeros((1152, nch))
        for k, a in (("gg", si.global_gain), ("sfs", si.scale_fac_scale), ("bt", si.block_type),
                     ("mixed", si.mixed_block_flag), ("pre", si.pre_flag), ("sbg", si.sub_block_gain),
                     ("sfl", si.scale_fac_l), ("sfsh", si.scale_fac_s)):
            rec[k].append(np.asarray(a).astype(np.int32).copy())
        rec["is"].append(isv)
        for gr in range(2):
            for ch in range(nch):
                RF.re_quantize(gr, ch, si.scale_fac_scale, si.block_type, si.mixed_block_flag,
                               hdr.band_width.short_win, si.global_gain, si.scale_fac_s, hdr.band_index.long_win,
                               si.scale_fac_l, si.pre_flag, fr._Frame__samples, si.sub_block_gain)
            if hdr.channel_mode == RF.ChannelMode.JointStereo and hdr.mode_extension[0]:
                fr._Frame__ms_stereo(gr)
            for ch in range(nch):
                if si.block_type[gr][ch] == 2 or si.mixed_block_flag[gr][ch]:
                    fr._Frame__reorder(gr, ch)
                else:
                    fr._Frame__alias_reduction(gr, ch)
                RF.imdct(gr, ch, si.block_type, fr._Frame__samples, fr._Frame__sine_block, fr._Frame__prev_samples)
                fr._Frame__frequency_inversion(gr, ch)
                RF.synth_filter_bank(gr, ch, fr._Frame__samples, fr._Frame__fifo, fr._Frame__synth_filter_bank_block)
        fr._Frame__interleave()
        rec["pcm"].append(fr.pcm.copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out["hdr"] = np.asarray(header_bytes(sr_code, mode, mode_ext), dtype=np.uint8)
    return out


def gen_stages():
    t0 = time.time()
    rng = np.random.default_rng(20250523)
    cases = {
        # name: (frames, sr_code, mode, mode_ext, block types, mixed?, max |is|)
        "long_44_stereo": (3, 0, 0, 0, [0], False, 8206),
        "all_44_ms": (4, 0, 1, 2, [0, 1, 2, 3], True, 2000),
        "all_48_stereo": (3, 1, 0, 0, [0, 1, 2, 3], True, 300),
        "all_32_ms": (3, 2, 1, 2, [0, 2, 2, 3], True, 300),
        "short_44_mono": (3, 0, 3, 0, [2, 0], False, 100),
        "joint_noms_44": (2, 0, 1, 1, [0, 2], False, 100),
    }
    out = {}
    for name, c in cases.items():
        r = run_chain(rng, *c)
        for k, v in r.items():
            out[f"{name}__{k}"] = v
    np.savez_compressed(os.path.join(HERE, "g4_decode_chain.npz"), **out)
    print("stages: decode chain", len(cases), "cases", f"{time.time() - t0:.1f}s")
    gen_enc_stages(rng)


# --------------------------------------------------------------------------
# G5  encoder stage vectors
# --------------------------------------------------------------------------
def gen_enc_stages(rng):
    t0 = time.time()
    wav = os.path.join(WORK, "tiny.wav")
    write_wav(wav, np.zeros((1152, 2), dtype=np.int16), 44100)
    enc = RE.MP3Encoder(RE.WavReader(wav, 128))
    out = {}
    # analysis filterbank: ring x + 32 new samples -> 32 subband values, 6 consecutive slots
    x = np.zeros((2, 512), dtype=np.int32)
    off = np.zeros(2, dtype=np.int32)
    pcm = rng.integers(-32768, 32768, size=(6, 32)).astype(np.int16)
    pcm[0, :4] = [-32768, 32767, -32768, 32767]
    sb = np.zeros((6, 32), dtype=np.int32)
    for k in range(6):
        for i in range(31, -1, -1):
            x[0][i + off[0]] = np.int32(pcm[k][31 - i]) << 16
        sb[k] = RE.window_filter_sub_band(np.zeros(32, dtype=np.int32), 0, x, off, enc._MP3Encoder__sub_band.fl)
    out["wf_pcm"] = pcm
    out["wf_sb"] = sb
    # quantize incl. the ln >= 10000 float path and the early-out
    l3 = enc._MP3Encoder__l3loop
    xr = (rng.standard_normal(576) * 10 ** rng.uniform(3, 9.3, size=576)).astype(np.int64)
    xr = np.clip(xr, -(2 ** 31 - 1), 2 ** 31 - 1).astype(np.int32)
    xrabs = np.abs(xr.astype(np.int64)).astype(np.int32)
    xrmax = int(xrabs.max())
    steps = list(range(-120, 1, 7))
    qix = np.zeros((len(steps), 576), dtype=np.int32)
    qmax = np.zeros(len(steps), dtype=np.int32)
    ix = np.zeros(576, dtype=np.int32)
    for n, st in enumerate(steps):
        qmax[n] = RE.quantize(ix, st, l3.steptabi, xrmax, xr, l3.int2idx, l3.steptab, xrabs)
        qix[n] = ix
    out["q_xr"] = xr
    out["q_steps"] = np.asarray(steps, dtype=np.int32)
    out["q_ix"] = qix
    out["q_max"] = qmax
    np.savez_compressed(os.path.join(HERE, "g5_encode_stages.npz"), **out)
    print("stages: encoder", f"{time.time() - t0:.1f}s")


# --------------------------------------------------------------------------
# G6  synthetic 44.1 kHz stereo stream @128 kbps with a hidden message
# --------------------------------------------------------------------------
def gen_synth():
    sys.path.insert(0, os.path.dirname(HERE))
    from synth_pcm import synth_pcm
    n_frames = 48
    pcm = synth_pcm(n_frames, seed=0x9E3779B97F4A7C15)
    # make the last 6 frames fade into digital silence: exercises xrmax == 0 and big_values == 0 (E7)
    fade = np.ones(n_frames * 1152)
    fade[-6 * 1152:-4 * 1152] = np.linspace(1, 0, 2 * 1152) ** 4
    fade[-4 * 1152:] = 0
    pcm = (pcm.astype(np.float64) * fade[:, None]).astype(np.int16)
    wav = os.path.join(WORK, "synth.wav")
    write_wav(wav, pcm, 44100)
    msg = "5#hello"
    hide = str_to_binary_str(msg)
    t0 = time.time()
    out, mp3 = encode_instrumented(wav, os.path.join(WORK, "synth.mp3"), 128, hide, keep_frames=None)
    out["hide_bits"] = np.frombuffer(hide.encode(), dtype=np.uint8) - ord("0")
    out["pcm"] = pcm
    out["mp3"] = np.frombuffer(mp3, dtype=np.uint8)
    print("synth encode:", len(mp3), f"{time.time() - t0:.1f}s")
    t0 = time.time()
    dec, _ = decode_instrumented(os.path.join(WORK, "synth.mp3"), os.path.join(WORK, "synth_dec.wav"), keep_pcm_frames=2)
    for k in ("bits", "pcm_sha256", "pcm_i16_sha256", "wav_sha256", "pcm_head", "is"):
        out["dec_" + k] = dec[k]
    np.savez_compressed(os.path.join(HERE, "g6_synth128.npz"), **out)
    print("synth decode:", len(dec["bits"]), "bits", f"{time.time() - t0:.1f}s")


# --------------------------------------------------------------------------
# G7  synthetic decode-only corpus: short/start/stop/mixed blocks, MS, mono, CRC, reservoir, ID3, books 4/14
# --------------------------------------------------------------------------
def gen_corpus():
    sys.path.insert(0, os.path.dirname(HERE))
    import frame_synth
    out = {}
    for name, data in frame_synth.corpus().items():
        t0 = time.time()
        mp3 = os.path.join(WORK, name + ".mp3")
        with open(mp3, "wb") as f:
            f.write(data)
        dec, pcm = decode_instrumented(mp3, os.path.join(WORK, name + ".wav"), keep_pcm_frames=2)
        out[name + "__mp3"] = np.frombuffer(data, dtype=np.uint8)
        for k in ("is", "bits", "kbps", "pcm_sha256", "pcm_i16_sha256", "wav_sha256", "pcm_head", "main_data_begin",
                  "frame_size", "table_select", "si_block_type", "si_mixed_block_flag", "scale_fac_l", "scale_fac_s"):
            out[name + "__" + k] = dec[k]
        out[name + "__pcm_rows"] = np.int64(pcm.shape[0])
        out[name + "__nch"] = np.int64(pcm.shape[1])
        print("corpus", name, "frames", dec["is"].shape[0], "rows", pcm.shape, "max|pcm|", float(np.abs(pcm).max()),
              f"{time.time() - t0:.1f}s")
    np.savez_compressed(os.path.join(HERE, "g7_decode_corpus.npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["tables", "decode", "encode"]
    for w in what:
        globals()["gen_" + w]()
