#!/usr/bin/env python3
"""Golden vectors for the ID3v2 listing the reference's Decoder writes to METADATA.txt when it is not quiet
(reference decoder/ID3_Parser.py, decoder/decoder.py:37-57).  Runs the upstream reference (build container only, via
refshim.py) on hand-made tags in front of a few real frames and records validity, audio offset, the METADATA.txt text,
or the exception the reference's constructor dies with.

    python tests/golden/gen_id3_golden.py        ->  tests/golden/g8_id3.json
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from refshim import load_reference  # noqa: E402

load_reference()
from mp3stego.decoder.decoder import Decoder as RDecoder  # noqa: E402


def syncsafe(n):
    return bytes([(n >> 21) & 0x7f, (n >> 14) & 0x7f, (n >> 7) & 0x7f, n & 0x7f])


def frame(fid, content, flags=0):
    return fid + syncsafe(len(content)) + bytes([flags >> 8, flags & 0xff]) + content


def tag(frames, flags=0, version=(3, 0), pad=0, size=None, ext=b""):
    body = ext + b"".join(frames) + bytes(pad)
    return b"ID3" + bytes(version) + bytes([flags]) + syncsafe(len(body) if size is None else size) + body


def cases():
    t = frame(b"TIT2", b"\x00A title")
    a = frame(b"TPE1", b"\x03Some artist \xc3\xa9")
    yield "none", b""
    yield "two_frames", tag([t, a])
    yield "padding", tag([t, a], pad=64)
    yield "v24", tag([t], version=(4, 0))
    yield "flag_unsync", tag([t], flags=0x80)
    yield "flag_extended", tag([t], flags=0x40, ext=b"")          # size field = the bytes where the first frame id sits
    yield "flag_extended_small", tag([t], flags=0x40, ext=syncsafe(6) + bytes(6))
    yield "flag_experimental", tag([t, a], flags=0x20)
    yield "flag_footer", tag([t], flags=0x10, pad=10)
    yield "flags_all", tag([t], flags=0xf0, ext=syncsafe(4) + bytes(4), pad=16)
    yield "protected_bit", tag([t], flags=0x01)
    yield "protected_bits", tag([t], flags=0x4c)
    for k, fl in enumerate((0x0001, 0x0002, 0x0004, 0x0100, 0x0200, 0x0400, 0x0707, 0x8080, 0xffff)):
        yield "frame_flags_%d" % k, tag([frame(b"TXXX", b"\x00k\x00v", fl), t])
    yield "not_utf8", tag([frame(b"APIC", b"\xff\xfe\x00\x01binary\x80\x81"), t])
    yield "empty_content", tag([frame(b"TCON", b""), t])
    yield "lowercase_id", tag([t, frame(b"Tabc", b"\x00x"), a])
    yield "digit_id", tag([frame(b"T123", b"\x00digits"), frame(b"1234", b"\x00all digits")])
    yield "latin_upper_id", tag([frame(b"\xc0\xc9\xd6\xdc", b"\x00upper case beyond ascii"), t])
    yield "space_in_id", tag([frame(b"TI 2", b"\x00x"), t])
    yield "oversized_frame", tag([t, b"TALB" + syncsafe(5000) + b"\x00\x00" + b"\x00short"], pad=8)
    yield "size_zero", tag([t], size=0)
    yield "size_smaller_than_frames", tag([t, a], size=len(t) + 3)
    yield "size_beyond_file", tag([t], size=20000)
    yield "id_cut_by_eof", None                                   # built below: the file ends inside a frame header


def main():
    g6 = np.load(os.path.join(HERE, "g6_synth128.npz"))["mp3"].tobytes()
    audio = g6[:3 * 418]
    out = []
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        for name, t in cases():
            if name == "id_cut_by_eof":
                data = tag([frame(b"TIT2", b"\x00A title")], size=400)[:10 + 13] + b"TP"
            elif name == "size_beyond_file":
                data = t + audio[:100]
            else:
                data = t + audio
            with open("case.mp3", "wb") as f:
                f.write(data)
            rec = {"name": name, "file_hex": data.hex(), "valid": None, "offset": None, "metadata": None, "raises": None}
            try:
                dec = RDecoder("case.mp3", "case.wav")
            except Exception as e:                                # noqa: BLE001 - whatever the reference dies with
                rec["raises"] = type(e).__name__
                out.append(rec)
                continue
            id3 = dec._Decoder__id3_decoder
            rec["valid"] = bool(id3.is_valid)
            if id3.is_valid:
                rec["offset"] = int(id3.offset)
                if os.path.exists("METADATA.txt"):
                    os.remove("METADATA.txt")
                dec._Decoder__parse_metadata(id3)
                rec["metadata"] = open("METADATA.txt", encoding="utf-8", errors="surrogateescape").read()
            out.append(rec)
        os.chdir(HERE)
    with open(os.path.join(HERE, "g8_id3.json"), "w") as f:
        json.dump(out, f, indent=1)
    for r in out:
        print(r["name"], r["valid"], r["offset"], r["raises"], None if r["metadata"] is None else len(r["metadata"]))


if __name__ == "__main__":
    main()
