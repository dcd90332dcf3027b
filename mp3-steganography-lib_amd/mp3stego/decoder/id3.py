"""The ID3v2 tag in front of the audio: where the audio starts, and the listing `Decoder.decode(quiet=False)` leaves
in METADATA.txt (reference decoder/ID3_Parser.py:85-193, decoder/decoder.py:37-57).

Host-side bookkeeping, nothing of it is on the device path (the library skips the tag by itself).  The reference's
arithmetic is kept as it is: size bytes are added unmasked, seven bits apart; the frame region is measured from the
tag's start but walked from behind its header, so it runs ten bytes past the tag; a frame id is legal when every
character is an upper-case letter or a digit in Python's sense of the words; flag names keep their spelling.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Union

TAG_FLAGS = ("FooterPresent", "ExperimentalIndicator", "ExtendedHeader", "Unsynchronisation")      # header bits 4..7
FRAME_FLAG_BITS = ((0, "DiscardFrameOnTagAlter"), (1, "DiscradFrameOnFileAlter"), (2, "ReadOnly"),
                   (8, "ZLIBCompression"), (9, "FrameEncrypted"), (10, "FrameContainsGroupInformation"))


@dataclass
class TagFrame:
    id: str
    raw: bytes
    flags: List[str] = field(default_factory=list)

    @property
    def content(self) -> Union[str, bytes]:
        try:
            return self.raw.decode("utf-8")
        except UnicodeDecodeError:
            return self.raw


@dataclass
class Tag:
    version: str
    flags: List[str]
    offset: int                       # first byte of audio
    frames: List[TagFrame]


def _size(four: bytes) -> int:
    if len(four) < 4:
        raise IndexError("ID3 size field cut short by the end of the file")      # the reference indexes past its slice
    n = 0
    for b in four:
        n = (n << 7) + b
    return n


def read_tag(data: bytes) -> Optional[Tag]:
    """the tag at the start of `data`, or None when there is none the reference accepts"""
    if len(data) < 3:
        raise IndexError("file shorter than an ID3 signature")        # the reference indexes bytes 0..2 unguarded
    if data[:3] != b"ID3":
        return None
    if len(data) < 6:
        raise IndexError("file ends inside the ID3 header")
    if data[5] & 0x0f:                                                # the four low flag bits must be clear
        return None
    set_bits = [bool(data[5] >> (4 + k) & 1) for k in range(4)]
    flags = [name for name, on in zip(TAG_FLAGS, set_bits) if on]
    offset = _size(data[6:10]) + (20 if set_bits[0] else 10)
    ext_field = _size(data[10:14])                                    # read (so it has to exist) with or without the flag
    ext = ext_field if set_bits[2] else 0
    start, span = 10 + ext, offset - ext - (10 if set_bits[0] else 0)
    frames, i = [], 0
    while i < span:
        fid = data[start + i:start + i + 4]
        name = fid.decode("latin-1")
        if not all(ch.isupper() or ch.isdigit() for ch in name):
            break
        size = _size(data[start + i + 4:start + i + 8])
        two = data[start + i + 8:start + i + 10].ljust(2, b"\0")      # flag bits past the end of the file read as 0
        word = two[0] << 8 | two[1]
        body = data[start + i + 10:start + i + 10 + size]
        frames.append(TagFrame(name, body, [label for bit, label in FRAME_FLAG_BITS if word >> bit & 1]))
        i += 10 + size
    return Tag(f"2.{data[3]}.{data[4]}", flags, offset, frames)


def listing(path: str, tag: Tag) -> str:
    """the text of METADATA.txt for the file at `path`"""
    out = [f"METADATA FOR FILE: {path}\n", "#" * 32 + "\n\n\n", f"ID3 Version: {tag.version}\n"]
    if tag.flags:
        out.append("ID3 Flags:\n" + "".join(f"- {f}\n" for f in tag.flags) + "\n")
    out.append("\nID3 Frames:\n")
    for k, fr in enumerate(tag.frames):
        out.append(f"Frame number: {k}\nFrame ID: {fr.id}\nContent: {fr.content}\n")
        if fr.flags:
            out.append("Frame Flags:\n" + "".join(f"- {f}\n" for f in fr.flags))
        out.append("\n")
    return "".join(out)
