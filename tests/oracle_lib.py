"""ctypes binding of oracle/liborc.so -- the CPU restatement used as the CHECKER.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORC_DIR, "liborc.so")


def build(force=False):
    srcs = [os.path.join(ORC_DIR, f) for f in os.listdir(ORC_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORC_DIR, "liborc.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class FrameInfo(C.Structure):
    _fields_ = [("hdr", C.c_int32 * 4), ("frame_size", C.c_int32), ("nch", C.c_int32), ("sr_idx", C.c_int32),
                ("main_data_begin", C.c_int32), ("scfsi", C.c_int32 * 8),
                ("part2_3_length", C.c_int32 * 4), ("big_value", C.c_int32 * 4), ("global_gain", C.c_int32 * 4),
                ("scale_fac_compress", C.c_int32 * 4), ("window_switching", C.c_int32 * 4),
                ("block_type", C.c_int32 * 4), ("mixed_block_flag", C.c_int32 * 4),
                ("region0_count", C.c_int32 * 4), ("region1_count", C.c_int32 * 4), ("pre_flag", C.c_int32 * 4),
                ("scale_fac_scale", C.c_int32 * 4), ("count1table_select", C.c_int32 * 4),
                ("table_select", C.c_int32 * 12), ("sub_block_gain", C.c_int32 * 12),
                ("scale_fac_l", C.c_int32 * 88), ("scale_fac_s", C.c_int32 * 156)]


FRAMEINFO_DTYPE = np.dtype([
    ("hdr", "<i4", (4,)), ("frame_size", "<i4"), ("nch", "<i4"), ("sr_idx", "<i4"), ("main_data_begin", "<i4"),
    ("scfsi", "<i4", (2, 4)),
    ("part2_3_length", "<i4", (2, 2)), ("big_value", "<i4", (2, 2)), ("global_gain", "<i4", (2, 2)),
    ("scale_fac_compress", "<i4", (2, 2)), ("window_switching", "<i4", (2, 2)), ("block_type", "<i4", (2, 2)),
    ("mixed_block_flag", "<i4", (2, 2)), ("region0_count", "<i4", (2, 2)), ("region1_count", "<i4", (2, 2)),
    ("pre_flag", "<i4", (2, 2)), ("scale_fac_scale", "<i4", (2, 2)), ("count1table_select", "<i4", (2, 2)),
    ("table_select", "<i4", (2, 2, 3)), ("sub_block_gain", "<i4", (2, 2, 3)),
    ("scale_fac_l", "<i4", (2, 2, 22)), ("scale_fac_s", "<i4", (2, 2, 3, 13))])

GI_FIELDS = ["part2_3_length", "big_values", "count1", "global_gain", "scale_fac_compress", "region0_count",
             "region1_count", "preflag", "scale_fac_scale", "count1table_select", "part2_length", "address1",
             "address2", "address3", "quantizerStepSize"]
GRINFO_DTYPE = np.dtype([(k, "<i4") for k in GI_FIELDS] + [("table_select", "<i4", (3,))])
ENCFRAME_DTYPE = np.dtype([("gi", GRINFO_DTYPE, (2, 2)), ("scfsi", "<i4", (2, 4)), ("written", "<i4"),
                           ("hide_off", "<i4"), ("padding", "<i4")])

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_long
    L.orc_dec_new.restype = vp
    L.orc_dec_free.argtypes = [vp]
    L.orc_dec_run.argtypes = [vp, vp, i64, i64]
    for n in ("n_frames", "n_pcm_rows", "n_bits"):
        getattr(L, "orc_dec_" + n).argtypes = [vp]
        getattr(L, "orc_dec_" + n).restype = i64
    for n in ("channels", "sampling_rate", "bit_rate"):
        getattr(L, "orc_dec_" + n).argtypes = [vp]
        getattr(L, "orc_dec_" + n).restype = i32
    for n in ("pcm", "bits", "is", "frames"):
        getattr(L, "orc_dec_" + n).argtypes = [vp]
        getattr(L, "orc_dec_" + n).restype = vp
    L.orc_enc_new.restype = vp
    L.orc_enc_new.argtypes = [i32, i32, i32, vp, i64]
    L.orc_enc_free.argtypes = [vp]
    L.orc_enc_run.argtypes = [vp, vp, i64]
    for n in ("n_frames", "out_len", "hide_offset"):
        getattr(L, "orc_enc_" + n).argtypes = [vp]
        getattr(L, "orc_enc_" + n).restype = i64
    for n in ("out", "frames", "mdct_freq", "ix"):
        getattr(L, "orc_enc_" + n).argtypes = [vp]
        getattr(L, "orc_enc_" + n).restype = vp
    L.orc_tables.restype = vp
    dp = np.ctypeslib.ndpointer(np.float64, flags="C")
    ip = np.ctypeslib.ndpointer(np.int32, flags="C")
    L.orc_requantize.argtypes = [dp, i32, i32, i32, i32, i32, ip, ip, ip, i32]
    L.orc_ms_stereo.argtypes = [dp, dp]
    L.orc_reorder.argtypes = [dp, i32]
    L.orc_alias_reduction.argtypes = [dp]
    L.orc_imdct.argtypes = [dp, i32, dp]
    L.orc_frequency_inversion.argtypes = [dp]
    L.orc_synth_filter_bank.argtypes = [dp, dp]
    L.orc_pcm_to_i16.argtypes = [C.c_double]
    L.orc_pcm_to_i16.restype = C.c_int16
    L.orc_enc_window_filter_subband.argtypes = [ip, ip, ip]
    L.orc_enc_quantize.argtypes = [ip, i32, i32, ip, ip]
    L.orc_enc_quantize.restype = i32
    L.orc_enc_rate_units.argtypes = [vp, i64, ip, ip, ip, vp, ip]
    L.orc_enc_rate_units.restype = None
    L.orc_enc_probe_bits.argtypes = [vp, i64, i32, ip, ip, ip, ip]
    L.orc_enc_probe_bits.restype = None
    _lib = L
    return L


def _arr(ptr, dtype, shape):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * n).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


def id3_offset(data: bytes) -> int:
    """decoder/ID3_Parser.py:95-131: offset of the first frame header (0 if no valid tag)."""
    if len(data) >= 10 and data[0:3] == b"ID3":
        flags = data[5]
        if flags & 0x0F:
            return 0
        size = 0
        for i in range(4):
            size = (size << 7) + data[6 + i]
        return size + (20 if (flags >> 4) & 1 else 10)
    return 0


def decode(data: bytes):
    """Run the oracle decoder on an MP3 file image; returns a dict of artefacts."""
    L = lib()
    d = L.orc_dec_new()
    try:
        buf = np.frombuffer(data, dtype=np.uint8)
        rc = L.orc_dec_run(d, buf.ctypes.data, len(data), id3_offset(data))
        nf, rows, nb = L.orc_dec_n_frames(d), L.orc_dec_n_pcm_rows(d), L.orc_dec_n_bits(d)
        nch = L.orc_dec_channels(d)
        out = {"rc": rc, "n_frames": nf, "channels": nch, "sampling_rate": L.orc_dec_sampling_rate(d),
               "bit_rate": L.orc_dec_bit_rate(d)}
        if rc == 0:
            out["pcm"] = _arr(L.orc_dec_pcm(d), np.float64, (rows, nch))
            out["bits"] = _arr(L.orc_dec_bits(d), np.uint8, (nb,))
            out["is"] = _arr(L.orc_dec_is(d), np.int16, (nf, 2, 2, 576))
            out["frames"] = _arr(L.orc_dec_frames(d), FRAMEINFO_DTYPE, (nf,))
        return out
    finally:
        L.orc_dec_free(d)


def pcm_to_i16(pcm: np.ndarray) -> np.ndarray:
    """(pcm * 32767).astype(int16) with numpy/x86 wrap semantics (MP3_Parser.py:91)."""
    x = pcm * 32767
    ok = np.abs(x) < 2147483648.0
    v = np.where(ok, x, 0.0).astype(np.int64)
    return (v & 0xFFFF).astype(np.uint16).view(np.int16)


def encode(pcm_i16: np.ndarray, samplerate: int, bitrate: int, hide_bits=None):
    """Run the oracle encoder on interleaved int16 PCM [n][nch]."""
    L = lib()
    pcm_i16 = np.ascontiguousarray(pcm_i16, dtype=np.int16)
    nch = 1 if pcm_i16.ndim == 1 else pcm_i16.shape[1]
    hb = None
    nh = 0
    if hide_bits is not None and len(hide_bits):
        hb = np.ascontiguousarray(hide_bits, dtype=np.uint8)
        nh = len(hb)
    e = L.orc_enc_new(samplerate, nch, bitrate, hb.ctypes.data if hb is not None else None, nh)
    try:
        rc = L.orc_enc_run(e, pcm_i16.ctypes.data, pcm_i16.shape[0])
        nf = L.orc_enc_n_frames(e)
        out = {"rc": rc, "n_frames": nf, "hide_offset": L.orc_enc_hide_offset(e)}
        out["mp3"] = _arr(L.orc_enc_out(e), np.uint8, (L.orc_enc_out_len(e),)).tobytes()
        out["frames"] = _arr(L.orc_enc_frames(e), ENCFRAME_DTYPE, (nf,))
        out["mdct_freq"] = _arr(L.orc_enc_mdct_freq(e), np.int32, (nf, 2, 2, 576))
        out["ix"] = _arr(L.orc_enc_ix(e), np.int32, (nf, 2, 2, 576))
        out["too_long"] = out["hide_offset"] < nh - 1
        return out
    finally:
        L.orc_enc_free(e)


def rate_units(samplerate: int, max_bits: np.ndarray, xr: np.ndarray, hide_bits=None):
    """The reference's rate loop (MP3_Encoder.py:766-813) on n spectra int32 [n][576], each as the first granule of a stream (fresh
    GrInfo; `hide_bits`: every unit sees the message from its start, cursor 0) with its own budget.  Returns {"ix": int32 [n][576] (unsigned, before format_bitstream), "gi": GRINFO_DTYPE [n],
    "rc": int32 [n]}."""
    L = lib()
    xr = np.ascontiguousarray(xr, dtype=np.int32).reshape(-1, 576)
    n = xr.shape[0]
    mb = np.ascontiguousarray(np.broadcast_to(np.asarray(max_bits, dtype=np.int32), (n,)))
    hb = None if hide_bits is None or len(hide_bits) == 0 else bytes(int(b) + 48 for b in hide_bits)
    e = L.orc_enc_new(samplerate, 2, 128, hb, 0 if hb is None else len(hb))
    try:
        ix = np.zeros((n, 576), dtype=np.int32)
        gi = np.zeros(n, dtype=GRINFO_DTYPE)
        rc = np.zeros(n, dtype=np.int32)
        L.orc_enc_rate_units(e, n, mb, xr, ix, gi.ctypes.data, rc)
    finally:
        L.orc_enc_free(e)
    return {"ix": ix, "gi": gi, "rc": rc}


def probe_bits(samplerate: int, step: int, xr: np.ndarray):
    """One probe of the reference's binary search (MP3_Encoder.py:973-990: quantize, then the loop body) at `step` on n spectra with
    fresh GrInfo: (bits, big_values, count1); bits 100000 where quantize refuses, -1 for silence or a step outside steptab."""
    L = lib()
    xr = np.ascontiguousarray(xr, dtype=np.int32).reshape(-1, 576)
    n = xr.shape[0]
    e = L.orc_enc_new(samplerate, 2, 128, None, 0)
    try:
        bits, bv, c1 = (np.zeros(n, dtype=np.int32) for _ in range(3))
        L.orc_enc_probe_bits(e, n, step, xr, bits, bv, c1)
    finally:
        L.orc_enc_free(e)
    return bits, bv, c1


def wav_bytes(pcm_i16: np.ndarray, rate: int) -> bytes:
    """scipy.io.wavfile.write layout for int16 data (44-byte header)."""
    import struct
    data = np.ascontiguousarray(pcm_i16, dtype="<i2")
    nch = 1 if data.ndim == 1 else data.shape[1]
    nb = data.nbytes
    return (b"RIFF" + struct.pack("<I", 36 + nb) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, nch, rate, rate * nch * 2, nch * 2, 16) +
            b"data" + struct.pack("<I", nb) + data.tobytes())
