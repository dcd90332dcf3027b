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


if __name__ == "__main__":
    what = sys.argv[1:] or ["tables", "decode", "encode"]
    for w in what:
        globals()["gen_" + w]()
