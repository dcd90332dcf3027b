"""MPEG-1 Layer III frame synthesiser for decode-side tests (SURVEY.md G7 / BASELINE config 5).

The reference encoder only ever writes long-block plain-stereo frames without bit reservoir, so the decode paths
for short / start / stop / mixed blocks, MS stereo, mono, CRC and main_data_begin > 0 need bit streams from
somewhere else.  This module writes syntactically valid frames with seeded random content: side info, scalefactors,
Huffman-coded big values (every code book incl. linbits), count1 quadruples (books A and B), and a real bit
reservoir.  It is a test tool: it uses the ISO code books from tests/golden/g1_tables.npz.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BITRATES = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
RATES = [44100, 48000, 32000]
SFB_LONG = {
    0: [0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576],
    1: [0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576],
    2: [0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576]}
SLEN = [[0, 0], [0, 1], [0, 2], [0, 3], [3, 0], [1, 1], [1, 2], [1, 3], [2, 1], [2, 2], [2, 3], [3, 1], [3, 2], [3, 3],
        [4, 2], [4, 3]]
LINBITS = [0] * 16 + [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]
XMAX = [0, 1, 2, 2, 0, 3, 3, 5, 5, 5, 7, 7, 7, 15, 0, 15] + [15] * 16   # largest base value per code book


class Bits:
    def __init__(self):
        self.b = []

    def put(self, val, n):
        for k in range(n - 1, -1, -1):
            self.b.append((int(val) >> k) & 1)

    def __len__(self):
        return len(self.b)

    def bytes(self):
        b = self.b + [0] * ((-len(self.b)) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


_books = None


def books():
    global _books
    if _books is None:
        g = np.load(os.path.join(GOLDEN, "g1_tables.npz"))
        _books = {}
        for t in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 24, 32, 33):
            _books[t] = (g[f"enc_hcod_{t}"].astype(np.int64), g[f"enc_hlen_{t}"].astype(np.int64))
        for t in range(17, 24):
            _books[t] = _books[16]
        for t in range(25, 32):
            _books[t] = _books[24]
    return _books


def put_pair(bw, t, x, y):
    """one big-value pair with code book t (reference decoder order: code, x linbits, x sign, y linbits, y sign)"""
    hc, hl = books()[t]
    ylen = {1: 2, 2: 3, 3: 3, 5: 4, 6: 4, 7: 6, 8: 6, 9: 6, 10: 8, 11: 8, 12: 8}.get(t, 16)
    ax, ay = abs(x), abs(y)
    bx, by = min(ax, 15) if t > 15 else ax, min(ay, 15) if t > 15 else ay
    bw.put(hc[bx * ylen + by], hl[bx * ylen + by])
    lb = LINBITS[t]
    if lb and bx == 15:
        bw.put(ax - 15, lb)
    if ax:
        bw.put(1 if x < 0 else 0, 1)
    if lb and by == 15:
        bw.put(ay - 15, lb)
    if ay:
        bw.put(1 if y < 0 else 0, 1)


def put_quad(bw, table_b, v):
    """count1 quadruple as the reference DEcoder reads it: book A code of (v,w,x,y) MSB first, or 4 inverted bits"""
    a = [abs(q) for q in v]
    idx = a[0] * 8 + a[1] * 4 + a[2] * 2 + a[3]
    if table_b:
        bw.put(15 - idx, 4)
    else:
        hc, hl = books()[32]
        bw.put(hc[idx], hl[idx])
    for q in v:
        if q:
            bw.put(1 if q < 0 else 0, 1)


def make_stream(seed, n_frames, sr_idx=0, bitrate_idx=9, mode=0, mode_ext=0, crc=False, block_types=(0,),
                allow_mixed=False, use_reservoir=True, tables=None, max_lin=40, id3=False):
    """Returns the bytes of an MP3 stream.  mode: 0 stereo, 1 joint, 3 mono."""
    rng = np.random.default_rng(seed)
    nch = 1 if mode == 3 else 2
    side_len = 17 if nch == 1 else 32
    tables = list(tables or [1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 19, 21, 23, 24, 26, 29, 31, 0])
    frames = []      # (header bytes, side-info writer fn needing main_data_begin, main data bytes, capacity)
    area_pos = data_pos = 0
    out_frames = []
    for f in range(n_frames):
        pad = int(rng.integers(0, 2)) if sr_idx == 0 else 0
        fsize = 144 * BITRATES[bitrate_idx] * 1000 // RATES[sr_idx] + pad
        cap = fsize - 4 - side_len - (2 if crc else 0)
        md = Bits()
        gran = [[None, None], [None, None]]
        scfsi = [[0] * 4 for _ in range(2)]
        mdb = area_pos - data_pos                         # bytes of reservoir in front of this frame's own area
        remaining = ((mdb if use_reservoir else 0) + cap) * 8 - 8
        want = int(cap * 8 * (0.55 + 0.8 * rng.random())) // (2 * nch)
        units_left = 2 * nch
        for gr in range(2):
            for ch in range(nch):
                bt = int(rng.choice(block_types))
                ws = bt != 0
                mixed = bool(ws and allow_mixed and rng.random() < 0.5)
                sfc = int(rng.integers(0, 16))
                sl0, sl1 = SLEN[sfc]
                bw = Bits()
                budget_bits = min(want, remaining // units_left, 4000)
                # ---- scalefactors, in the order the reference reads them (Frame.py:365-441)
                if bt == 2 and ws:
                    if mixed:
                        for _ in range(8):
                            bw.put(rng.integers(0, 1 << sl0) if sl0 else 0, sl0)
                        for _ in range(3 * 3):
                            bw.put(rng.integers(0, 1 << sl0) if sl0 else 0, sl0)
                    else:
                        for _ in range(6 * 3):
                            bw.put(rng.integers(0, 1 << sl0) if sl0 else 0, sl0)
                    for _ in range(6 * 3):
                        bw.put(rng.integers(0, 1 << sl1) if sl1 else 0, sl1)
                elif gr == 0:
                    for s in range(21):
                        sl = sl0 if s < 11 else sl1
                        bw.put(rng.integers(0, 1 << sl) if sl else 0, sl)
                else:
                    g0 = gran[0][ch]
                    g0_short = g0["bt"] == 2 and g0["ws"]
                    for b in range(4):
                        scfsi[ch][b] = int((not g0_short) and rng.random() < 0.4)
                    for s in range(21):
                        band = 0 if s < 6 else (1 if s < 11 else (2 if s < 16 else 3))
                        sl = sl0 if s < 11 else sl1
                        if not scfsi[ch][band]:
                            bw.put(rng.integers(0, 1 << sl) if sl else 0, sl)
                # ---- regions
                if ws:
                    r0c, r1c = (8 if bt == 2 else 7), (12 if bt == 2 else 13)
                    if bt == 2:
                        reg0, reg1 = 36, 576
                    else:
                        reg0, reg1 = SFB_LONG[sr_idx][r0c + 1], SFB_LONG[sr_idx][r0c + 1 + r1c + 1]
                    ts = [int(rng.choice(tables)), int(rng.choice(tables)), 0]
                else:
                    r0c = int(rng.integers(0, 16))
                    r1c = int(rng.integers(0, min(8, 21 - r0c)))
                    reg0, reg1 = SFB_LONG[sr_idx][r0c + 1], SFB_LONG[sr_idx][r0c + 1 + r1c + 1]
                    ts = [int(rng.choice(tables)) for _ in range(3)]
                # ---- big values until the bit budget is used
                big_values = 0
                target_pairs = int(rng.integers(0, 289))
                for p in range(target_pairs):
                    i = 2 * p
                    t = ts[0] if i < reg0 else (ts[1] if i < reg1 else ts[2])
                    if len(bw) > budget_bits - 60:
                        break
                    if t not in (0, 4, 14):
                        lim = XMAX[t]
                        x, y = int(rng.integers(0, lim + 1)), int(rng.integers(0, lim + 1))
                        if LINBITS[t]:
                            if x == 15:
                                x += int(rng.integers(0, min(max_lin, (1 << LINBITS[t]) - 1) + 1))
                            if y == 15:
                                y += int(rng.integers(0, min(max_lin, (1 << LINBITS[t]) - 1) + 1))
                        x *= int(rng.choice([-1, 1]))
                        y *= int(rng.choice([-1, 1]))
                        put_pair(bw, t, x, y)
                    big_values = p + 1
                # ---- count1 quadruples
                c1b = int(rng.integers(0, 2))
                nq = int(rng.integers(0, 40))
                line = 2 * big_values
                for _ in range(nq):
                    if line + 4 >= 576 or len(bw) > budget_bits - 12:
                        break
                    put_quad(bw, c1b, [int(rng.integers(0, 2)) * int(rng.choice([-1, 1])) for _ in range(4)])
                    line += 4
                p23 = len(bw)
                if rng.random() < 0.3 and p23 + 8 <= remaining // units_left:
                    p23 += int(rng.integers(0, 9))          # slack inside part2_3_length: read as further quadruples
                    bw.put(0, p23 - len(bw))
                remaining -= p23
                units_left -= 1
                gran[gr][ch] = dict(p23=p23, bv=big_values, gg=int(rng.integers(120, 200)), sfc=sfc, ws=ws, bt=bt,
                                    mixed=mixed, ts=ts, r0c=r0c, r1c=r1c, pre=int(rng.integers(0, 2)),
                                    sfs=int(rng.integers(0, 2)), c1=c1b, sbg=[int(v) for v in rng.integers(0, 8, 3)])
                md.b += bw.b
        md_bytes = md.bytes()
        # ---- bit reservoir: this frame's data starts mdb bytes before its own area
        if not use_reservoir:
            md_bytes += bytes(cap - len(md_bytes))
        elif mdb + cap - len(md_bytes) > 500:                 # keep main_data_begin <= 511: stuff this frame
            md_bytes += bytes(mdb + cap - len(md_bytes) - 500)
        assert len(md_bytes) <= mdb + cap, (f, len(md_bytes), mdb, cap)
        hdr = [0xFF, 0xFA | (0 if crc else 1), (bitrate_idx << 4) | (sr_idx << 2) | (pad << 1), (mode << 6) | (mode_ext << 4)]
        si = Bits()
        si.put(mdb, 9)
        si.put(0, 5 if nch == 1 else 3)
        for ch in range(nch):
            for b in range(4):
                si.put(scfsi[ch][b], 1)
        for gr in range(2):
            for ch in range(nch):
                g = gran[gr][ch]
                si.put(g["p23"], 12); si.put(g["bv"], 9); si.put(g["gg"], 8); si.put(g["sfc"], 4); si.put(int(g["ws"]), 1)
                if g["ws"]:
                    si.put(g["bt"], 2); si.put(int(g["mixed"]), 1)
                    si.put(g["ts"][0], 5); si.put(g["ts"][1], 5)
                    for w in range(3):
                        si.put(g["sbg"][w], 3)
                else:
                    for r in range(3):
                        si.put(g["ts"][r], 5)
                    si.put(g["r0c"], 4); si.put(g["r1c"], 3)
                si.put(g["pre"], 1); si.put(g["sfs"], 1); si.put(g["c1"], 1)
        assert len(si) == side_len * 8
        out_frames.append((bytes(hdr) + (b"\xAB\xCD" if crc else b"") + si.bytes(), cap))
        frames.append(md_bytes)
        data_pos += len(md_bytes)
        area_pos += cap
    # ---- lay the main data out in the frames' areas
    stream_md = b"".join(frames)
    stream_md += bytes(max(0, area_pos - len(stream_md)))
    out = bytearray()
    pos = 0
    for hdr_side, cap in out_frames:
        out += hdr_side + stream_md[pos:pos + cap]
        pos += cap
    if id3:
        out = bytearray(b"ID3\x03\x00\x00" + bytes([0, 0, 0, 23]) + b"TIT2" + bytes([0, 0, 0, 13, 0, 0]) + b"\x00synth title\x00") + out
    return bytes(out)


CORPUS = {
    # name: kwargs of make_stream
    "long_reservoir_44": dict(seed=11, n_frames=8, sr_idx=0, bitrate_idx=9, mode=0),
    "joint_ms_blocks_48": dict(seed=12, n_frames=8, sr_idx=1, bitrate_idx=11, mode=1, mode_ext=2, block_types=(0, 1, 2, 3)),
    "mono_crc_32": dict(seed=13, n_frames=8, sr_idx=2, bitrate_idx=5, mode=3, crc=True, block_types=(0, 2)),
    "mixed_blocks_44": dict(seed=14, n_frames=6, sr_idx=0, bitrate_idx=12, mode=0, block_types=(0, 2, 1, 3), allow_mixed=True),
    "books_4_14_id3": dict(seed=15, n_frames=6, sr_idx=0, bitrate_idx=14, mode=1, mode_ext=0, tables=[4, 14, 13, 24, 31, 0, 7],
                           max_lin=8191, id3=True),
    "no_reservoir_32k_lowrate": dict(seed=16, n_frames=6, sr_idx=2, bitrate_idx=1, mode=0, use_reservoir=False, block_types=(0, 2)),
}


def corpus():
    return {k: make_stream(**v) for k, v in CORPUS.items()}
