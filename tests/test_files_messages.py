"""Container formats, message framing and the byte-string facade of include/mp3s.h section (vi).
CPU part: WAV header in/out, framing and the reveal parse against Python restatements of the reference lines they
replace; GPU part: whole files through mp3s_decode_file / encode_file / hide_message / clear_file against the
reference's known answers (tests/golden/g3_facade.json)."""
import hashlib
import io
import json
import os
import struct

import numpy as np
import pytest


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def wav_of(pcm_i16, rate):
    from scipy.io import wavfile
    f = io.BytesIO()
    wavfile.write(f, rate, pcm_i16)
    return f.getvalue()


def py_reveal(bits):
    """reference decoder/decoder.py:90-108, restated on a '0'/'1' string"""
    bits = "".join("1" if b else "0" for b in bits)
    output_str = ''.join(chr(int(''.join(x), 2)) for x in zip(*[iter(bits)] * 8))
    message_len_str = ''
    for ch in output_str:
        if ch == '#':
            break
        message_len_str += ch
    try:
        message_len = int(message_len_str)
    except Exception:
        message_len = 0
        message_len_str = ""
    if (len(message_len_str) + 1 + message_len) > len(output_str):
        output_str = output_str[len(message_len_str) + 1:]
    else:
        output_str = output_str[len(message_len_str) + 1: len(message_len_str) + 1 + message_len]
    return bytes(output_str, 'utf-8')


# ------------------------------------------------------------------------------------------------ CPU: WAV
def test_wav_header_is_scipys(mlib):
    for rows, nch, rate in ((0, 2, 44100), (1152, 2, 48000), (41472, 2, 44100), (2304, 1, 32000)):
        pcm = np.zeros((rows, nch) if nch == 2 else (rows,), dtype=np.int16)
        assert mlib.wav_header(rows, nch, rate) == wav_of(pcm, rate)[:44]


def test_wav_parse_fields_and_quirks(mlib):
    pcm = (np.arange(2 * 1152 * 2, dtype=np.int32) % 65536 - 32768).astype(np.int16).reshape(-1, 2)
    w = wav_of(pcm, 44100)
    r = mlib.wav_parse(w, 128)
    assert (r["channels"], r["samplerate"], r["bits_per_sample"], r["bitrate"]) == (2, 44100, 16, 128)
    assert r["num_of_samples"] == 2304 and r["data_offset"] == 44
    assert r["n_values"] == 4608                      # np.fromfile asks for twice the values; the file ends first
    r = mlib.wav_parse(w + b"\0" * 100000, 128)
    assert r["n_values"] == 2 * 4608                  # ... and gets them when trailing bytes exist
    # a junk prefix: every tag is searched in the first 128 bytes, wherever it sits
    r = mlib.wav_parse(b"junk" * 5 + w, 320)
    assert r["data_offset"] == 64 and r["num_of_samples"] == 2304
    # 8-bit header: the sample count follows the header, the samples are still read as int16
    w8 = bytearray(w); w8[34:36] = struct.pack("<H", 8)
    assert mlib.wav_parse(bytes(w8), 320)["num_of_samples"] == 4608
    # odd chunk size: float arithmetic, truncated
    wo = bytearray(w); wo[40:44] = struct.pack("<I", 4609 * 2 - 1)
    assert mlib.wav_parse(bytes(wo), 320)["num_of_samples"] == int((4609 * 2 - 1) * 8 / 16 / 2)


def test_wav_parse_rejections_carry_the_reference_text(mlib):
    w = bytearray(wav_of(np.zeros((1152, 2), dtype=np.int16), 44100))

    def patched(at, data):
        x = bytearray(w); x[at:at + len(data)] = data
        return bytes(x)
    cases = [
        (b"", 'Bad WAVE file.'),
        (patched(0, b"RIFX"), 'Bad WAVE file.'),
        (patched(8, b"WAVX"), 'Bad WAVE file.'),
        (patched(12, b"fmtx"), 'Bad WAVE file.'),
        (patched(36, b"dat_"), 'Bad WAVE file.'),
        (bytes(w[:36]) + b"\0" * 200, 'Bad WAVE file.'),                          # no data tag in the first 128 bytes
        (patched(16, struct.pack("<I", 18)), 'Unsupported WAVE file, compression used instead of PCM.'),
        (patched(20, struct.pack("<H", 3)), 'Unsupported WAVE file, compression used instead of PCM.'),
        (patched(24, struct.pack("<I", 22050)), 'Unsupported sampling frequency.'),
        (patched(34, struct.pack("<H", 24)), 'Unsupported WAVE file, samples not int8, int16 or int32 type.'),
    ]
    for data, text in cases:
        with pytest.raises(mlib.Mp3sError) as e:
            mlib.wav_parse(data, 320)
        assert e.value.code == mlib.E_EXIT and e.value.text == text, text
    for kbps in (0, 100, 321, -2):
        with pytest.raises(mlib.Mp3sError) as e:
            mlib.wav_parse(bytes(w), kbps)
        assert e.value.code == mlib.E_EXIT and e.value.text == "Unsupported bitrate configuration."
    for kbps in (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, -1):   # -1 is in the reference's table
        assert mlib.wav_parse(bytes(w), kbps)["bitrate"] == kbps
    # where the reference dies in struct.unpack / ZeroDivisionError the library reports malformed input
    for data in (b"RIFF", bytes(w[:20]), bytes(w[:35]), patched(22, struct.pack("<H", 0))):
        with pytest.raises(mlib.Mp3sError) as e:
            mlib.wav_parse(data, 320)
        assert e.value.code == mlib.E_MALFORMED


# ------------------------------------------------------------------------------------------------ CPU: messages
def test_message_frame_is_the_references_bit_string(mlib):
    for m in ("ddd", "", "a#b", "ddd" * 100, "héllo wörld", "日本語", "x" * 1000):
        framed = str(len(m)) + "#" + m                              # steganography.py:45
        expect = "".join(format(b, "08b") for b in framed.encode("utf-8"))   # str_to_binary_str
        got = mlib.message_frame(m)
        assert "".join("1" if b else "0" for b in got) == expect, m


def test_message_reveal_matches_reference_parse(mlib):
    rng = np.random.default_rng(7)
    texts = [b"3#abcdef", b"abc", b" 2 #xyz", b"1_0#0123456789abc", b"-2#abcdef", b"99#ab", b"#abc", b"\xe9#ab",
             b"2#\xe9\xe8z", b"+1#ab", b"1__0#abc", b"_1#abc", b"1_#abc", b"-100#abcdef", b"\x1f3\xa0#abcd", b"0#", b"",
             b"5#\xff\x80\x7f\x00#", b"12345678901234567890123#ab", b"-12345678901234567890123#ab", b"3#ab", b"+#ab",
             b"- 1#ab", b"\t4\n#abcdefgh"]
    for t in texts:
        bits = np.unpackbits(np.frombuffer(t, dtype=np.uint8))
        assert mlib.message_reveal(bits) == py_reveal(bits), t
        for extra in (1, 7):                                        # an incomplete last byte is dropped
            b2 = np.concatenate([bits, np.ones(extra, dtype=np.uint8)])
            assert mlib.message_reveal(b2) == py_reveal(b2), t
    alphabet = np.frombuffer(b"0123456789#_+- \xa0\x1c\x85\nab\xe9\x00", dtype=np.uint8)
    for _ in range(3000):
        t = bytes(rng.choice(alphabet, size=int(rng.integers(0, 12))))
        bits = np.unpackbits(np.frombuffer(t, dtype=np.uint8)) if t else np.zeros(0, dtype=np.uint8)
        assert mlib.message_reveal(bits) == py_reveal(bits), t
    for _ in range(300):
        bits = rng.integers(0, 2, size=int(rng.integers(0, 200))).astype(np.uint8)
        assert mlib.message_reveal(bits) == py_reveal(bits)


def test_frame_then_reveal_round_trip_and_its_utf8_quirk(mlib):
    assert mlib.message_reveal(mlib.message_frame("hello")) == b"hello"
    # the count is in characters, the payload in UTF-8 bytes read back one chr() per byte: a non-ASCII message comes back
    # cut short and double-encoded, exactly as the reference reveals it
    assert mlib.message_reveal(mlib.message_frame("héllo")) == "hÃ©ll".encode("utf-8")


def test_reveal_message_needs_no_device(mlib, golden_dir):
    facade = json.load(open(os.path.join(golden_dir, "g3_facade.json")))
    with open(os.path.join(golden_dir, "g3_hide_ddd.mp3"), "rb") as f:
        r = mlib.reveal_message(f.read())
    assert r["data"].decode() == facade["revealed"] and r["kbps"] == 320 and r["channels"] == 2
    with open(os.path.join(golden_dir, "test.mp3"), "rb") as f:
        data = f.read()
    r = mlib.reveal_message(data)
    assert r["data"] == py_reveal(mlib.parse_stream(data)["bits"]) and r["n_frames"] == 36
    with pytest.raises(mlib.Mp3sError) as e:
        mlib.reveal_message(b"\xff\xfb")                                 # a sync and nothing behind it
    assert e.value.code == mlib.E_MALFORMED
    r = mlib.reveal_message(b"\x00" * 500)                              # no sync at all: nothing parsed, nothing hidden
    assert r["data"] == b"" and r["n_frames"] == 0 and r["kbps"] == 0


# ------------------------------------------------------------------------------------------------ GPU: whole files
@pytest.mark.gpu
def test_files_through_the_library_match_the_reference(ctx, mlib, golden_dir):
    facade = json.load(open(os.path.join(golden_dir, "g3_facade.json")))
    with open(os.path.join(golden_dir, "test.mp3"), "rb") as f:
        mp3 = f.read()
    d = ctx.decode_file(mp3)
    assert sha(d["data"]) == "d6787aaa" + sha(d["data"])[8:] and len(d["data"]) == 165932 and d["kbps"] == 320
    w = np.load(os.path.join(golden_dir, "g3_testmp3_wav_pcm.npz"))
    assert d["data"] == wav_of(w["pcm"], int(w["rate"]))            # scipy's bytes for the reference's PCM
    plain = ctx.encode_file(d["data"], 320)
    assert sha(plain["data"]).startswith("bb7a68cc") and len(plain["data"]) == 37616
    hid = ctx.hide_message(mp3, "ddd")
    assert sha(hid["data"]) == facade["hide_sha256"] and hid["too_long"] is facade["too_long"]
    assert mlib.reveal_message(hid["data"])["data"].decode() == facade["revealed"]
    long = ctx.hide_message(mp3, "ddd" * 100)
    assert sha(long["data"]) == facade["hide_long_sha256"] and long["too_long"] is facade["too_long_300"]
    assert mlib.reveal_message(long["data"])["data"].decode() == facade["revealed_long"]
    cleared = ctx.clear_file(hid["data"])
    assert sha(cleared["data"]) == facade["cleared_sha256"]
    assert mlib.reveal_message(cleared["data"])["data"].decode() == facade["revealed_cleared"]
    # the fused path (PCM stays in HBM) equals decode_file -> encode_file through host memory
    bits = mlib.message_frame("ddd")
    assert ctx.encode_file(d["data"], d["kbps"], bits)["data"] == hid["data"]


@pytest.mark.gpu
def test_reencode_of_corpus_streams(ctx, mlib, orc, golden_dir):
    """hide/clear on streams the reference encoder cannot make (reservoir, short blocks, CRC, ID3, a bad trailing
    header that duplicates the last frame): decode on the device, re-encode from HBM, compare with the oracle chain"""
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    done = 0
    for n in names:
        data = g[n + "__mp3"].tobytes()
        od = orc.decode(data)
        kbps = od["bit_rate"] // 1000
        if int(g[n + "__nch"]) != 2:
            with pytest.raises(mlib.Mp3sError) as e:
                ctx.hide_message(data, "x")
            assert e.value.code == mlib.E_UNSUPPORTED        # mono: IndexError in the reference encoder
            continue
        pcm16 = orc.pcm_to_i16(od["pcm"])
        msg = mlib.message_frame("corpus " + n)
        exp = orc.encode(pcm16, od["sampling_rate"], kbps, msg)
        got = ctx.hide_message(data, "corpus " + n)
        assert exp["rc"] == 0 and got["data"] == exp["mp3"], n
        assert got["hide_offset"] == exp["hide_offset"] and got["too_long"] == bool(exp["too_long"])
        assert ctx.clear_file(data)["data"] == orc.encode(pcm16, od["sampling_rate"], kbps, None)["mp3"], n
        done += 1
    assert done >= 3


@pytest.mark.gpu
def test_encode_file_sample_count_rules(ctx, mlib):
    ok = wav_of(np.zeros((2 * 1152, 2), dtype=np.int16), 44100)
    assert ctx.encode_file(ok, 128)["n_frames"] == 2
    for bad in (wav_of(np.zeros((1152 + 7, 2), dtype=np.int16), 44100),      # partial last frame
                wav_of(np.zeros(2304, dtype=np.int16), 44100),               # mono
                ok[:-10]):                                                   # data chunk longer than the file
        with pytest.raises(mlib.Mp3sError) as e:
            ctx.encode_file(bad, 128)
        assert e.value.code == mlib.E_UNSUPPORTED
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.encode_file(ok, 100)
    assert e.value.code == mlib.E_EXIT and e.value.text == "Unsupported bitrate configuration."
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.encode_file(ok, -1)
    assert e.value.code == mlib.E_UNSUPPORTED


def test_id3_listing_matches_reference(golden_dir):
    """METADATA.txt of a non-quiet decode (reference decoder/ID3_Parser.py, decoder.py:37-57): validity, audio offset and
    the listing text for hand-made tags, against what the reference made of them (tests/golden/gen_id3_golden.py)"""
    import json
    from mp3stego.decoder import id3
    cases = json.load(open(os.path.join(golden_dir, "g8_id3.json")))
    assert len(cases) >= 30
    for r in cases:
        data = bytes.fromhex(r["file_hex"])
        if r["raises"] and r["name"] == "id_cut_by_eof":             # the reference's ID3 constructor dies on it
            with pytest.raises(IndexError):
                id3.read_tag(data)
            continue
        tag = id3.read_tag(data)
        if r["raises"]:                                              # (dies later, in the frame parser: offset past the file)
            assert tag is not None and tag.offset > len(data)
            continue
        assert (tag is not None) == r["valid"], r["name"]
        if tag is not None:
            assert tag.offset == r["offset"], r["name"]
            assert id3.listing("case.mp3", tag) == r["metadata"], r["name"]
    with pytest.raises(IndexError):
        id3.read_tag(b"ID")


def test_scan_skips_the_tag_the_listing_describes(mlib, golden_dir):
    """the library's own tag skip and the listing agree on where the audio starts"""
    import json
    from mp3stego.decoder import id3
    for r in json.load(open(os.path.join(golden_dir, "g8_id3.json"))):
        if r["raises"] or not r["valid"]:
            continue
        data = bytes.fromhex(r["file_hex"])
        tag = id3.read_tag(data)
        s = mlib.scan_stream(data)
        plain = mlib.scan_stream(data[tag.offset:])
        assert s["n_frames"] == plain["n_frames"] and np.array_equal(s["frame_size"], plain["frame_size"]), r["name"]


@pytest.mark.gpu
def test_encode_file_takes_the_frame_behind_a_partial_one_like_the_reference(ctx, mlib, golden_dir):
    """a WAV whose declared sample count is not a multiple of 1152 per channel: the reference encodes one more frame from
    whatever follows in its buffer (here a LIST chunk behind the data chunk), and raises IndexError when the file ends
    inside that frame (tests/golden/gen_wav_tail_golden.py ran the reference on both files)"""
    g = np.load(os.path.join(golden_dir, "g9_wav_tail.npz"))
    assert str(g["tail_error"]) == "" and str(g["short_error"]) == "IndexError"
    r = ctx.encode_file(g["tail_wav"].tobytes(), 128)
    assert r["n_frames"] == 3 and bytes(r["data"]) == g["tail_mp3"].tobytes()
    with pytest.raises(mlib.Mp3sError) as e:
        ctx.encode_file(g["short_wav"].tobytes(), 128)
    assert e.value.code == mlib.E_UNSUPPORTED
