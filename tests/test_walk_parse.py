"""The byte-level scan split in two: a walk from frame header to frame header on the
host (mp3s_walk_stream) and the side-info parse + main-data gather on the device (mp3s_parse_frames_dev).  Reference:
decoder/MP3_Parser.py:68-80, Frame.py:288-363, FrameSideInformation.py:39-137.  The checker is the host scan
(mp3s_scan_stream), itself pinned to the reference by tests/test_decode_corpus.py and tests/test_fuzz.py."""
import os

import numpy as np
import pytest

from test_fuzz import header_mutants, mutants


def streams(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    out = {k.split("__")[0]: g[k].tobytes() for k in g.files if k.endswith("__mp3")}
    out["test.mp3"] = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    out["hide_ddd"] = open(os.path.join(golden_dir, "g3_hide_ddd.mp3"), "rb").read()
    return out


def check_walk(mlib, data):
    """the walk says `regular` only for streams it reproduces the scan on, frame by frame; -> (walk, scan) or None"""
    try:
        w = mlib.walk_stream(data)
    except mlib.Mp3sError:
        with pytest.raises(mlib.Mp3sError):          # refused before the first frame: by both
            mlib.scan_stream(data)
        return None
    try:
        s = mlib.scan_stream(data)
    except mlib.Mp3sError:
        assert not w["regular"]                      # the scan fails somewhere inside: the walk must not vouch for it
        return None
    if not w["regular"]:
        return None
    assert w["n_frames"] == s["n_frames"] and w["channels"] == s["channels"]
    assert w["sampling_rate"] == s["sampling_rate"] and w["bit_rate"] == s["bit_rate"] and w["dup_last_frame"] == s["dup_last_frame"]
    assert np.array_equal(w["refs"]["frame_size"], s["frame_size"])
    assert np.array_equal(w["refs"]["md_off"], s["side"]["md_off"]) and np.array_equal(w["refs"]["md_len"], s["side"]["md_len"])
    assert w["blob_len"] == len(s["blob"]) and w["max_part2_3_length"] == s["max_part2_3_length"]
    if w["n_frames"]:
        starts = np.concatenate([[0], np.cumsum(s["frame_size"][:-1])]) + int(w["refs"]["file_off"][0])
        assert np.array_equal(w["refs"]["file_off"], starts)
    if s["channels"] == 2 and w["n_frames"]:
        u = s["side"]["unit"]                        # [n][gr][ch]
        nz = (u["table_select"] != 0)
        nz[..., 2] &= u["window_switching"] == 0
        tables = nz.sum(axis=-1) * (u["big_values"] != 0)
        assert np.array_equal(w["tables"], tables.transpose(0, 2, 1).reshape(-1, 4))      # encoder order: (ch, gr)
        assert w["any_silent"] == bool((tables == 0).any())
    return w, s


def test_walk_equals_scan_on_corpus(mlib, golden_dir):
    regular = 0
    for name, data in streams(golden_dir).items():
        r = check_walk(mlib, data)
        assert r is not None, name                   # every stream of the corpus is a regular MPEG-1 Layer III stream
        regular += 1
    assert regular >= 8


def test_walk_on_cut_and_tagged_files(mlib, golden_dir):
    data = streams(golden_dir)["test.mp3"]
    # truncated inside the last frame / inside a header / inside the side info; an ID3v2 tag in front; garbage behind
    for cut in (1, 3, 5, 17, 40, 200, 1044, 1045, 1046):
        check_walk(mlib, data[:-cut])
    tag = b"ID3\x03\x00\x00\x00\x00\x02\x01" + bytes(257)
    r = check_walk(mlib, tag + data)
    assert r is not None and int(r[0]["refs"]["file_off"][0]) == len(tag)
    r = check_walk(mlib, data + b"\x00" * 700)       # a bad header ends the stream, the last frame is repeated (D12)
    assert r is not None and r[0]["dup_last_frame"] == 1
    # a stream cut in the middle: its first frames point in front of the file -- not the walk's business
    res = streams(golden_dir)
    name = next(n for n in res if "reservoir" in n or "resv" in n) if any(("reservoir" in n or "resv" in n) for n in res) else None
    if name:
        s = mlib.scan_stream(res[name])
        first = len(res[name]) - int(s["frame_size"].sum())
        cut = first + int(s["frame_size"][:3].sum())
        check_walk(mlib, res[name][cut:])


def test_walk_never_vouches_for_a_mutant_it_gets_wrong(mlib, golden_dir):
    data = streams(golden_dir)
    n_regular = n_all = 0
    srcs = [("test.mp3", 11), ("hide_ddd", 12)] + [(n, 13 + i) for i, n in enumerate(sorted(data)) if n not in ("test.mp3", "hide_ddd")]
    for src, seed in srcs:
        for m in list(header_mutants(mlib, data[src], 300, seed)) + list(mutants(data[src], 200, seed + 100)):
            n_all += 1
            n_regular += check_walk(mlib, m) is not None
    assert n_regular > n_all // 4                    # most damage leaves a stream the walk still takes


def parse_and_compare(ctx, mlib, data, w, s):
    d = ctx.parse_frames(data, w)
    assert (d["status"] & mlib.PS_MISMATCH) == 0
    assert bool(d["status"] & mlib.PS_INHERITS) == (not s["gpu_ok"])
    assert np.array_equal(d["blob"][:len(s["blob"])], s["blob"])
    assert np.array_equal(d["hdr"], s["hdr"])
    side, ref = d["side"].copy(), s["side"].copy()
    # the two fields a granule keeps from the frame before instead of parsing them: zero on the device
    ws = ref["unit"]["window_switching"] != 0
    ref["unit"]["table_select"][..., 2][ws] = 0
    ref["unit"]["sub_block_gain"][~ws] = 0
    assert side.tobytes() == ref.tobytes()
    bits, _ = mlib.stego_bits(d["tsel"], w["channels"])
    assert np.array_equal(bits, s["bits"])


@pytest.mark.gpu
def test_device_parse_equals_scan_on_corpus(ctx, mlib, golden_dir):
    for name, data in streams(golden_dir).items():
        r = check_walk(mlib, data)
        assert r is not None, name
        if r[0]["n_frames"]:
            parse_and_compare(ctx, mlib, data, *r)


@pytest.mark.gpu
def test_device_parse_on_cut_tagged_and_damaged_files(ctx, mlib, golden_dir):
    data = streams(golden_dir)
    base = data["test.mp3"]
    cases = [base[:-c] for c in (1, 3, 17, 40, 200, 1044)] + [b"ID3\x03\x00\x00\x00\x00\x02\x01" + bytes(257) + base, base + b"\x00" * 700]
    srcs = [("test.mp3", 21), ("hide_ddd", 22)] + [(n, 23 + i) for i, n in enumerate(sorted(data)) if n not in ("test.mp3", "hide_ddd")]
    for src, seed in srcs:
        cases += list(header_mutants(mlib, data[src], 100, seed)) + list(mutants(data[src], 60, seed + 100))
    checked = 0
    for m in cases:
        r = check_walk(mlib, m)
        if r is not None and r[0]["n_frames"]:
            parse_and_compare(ctx, mlib, m, *r)
            checked += 1
    assert checked > len(cases) // 4


@pytest.mark.gpu
def test_device_parse_then_huffman_equals_host_parser(ctx, mlib, golden_dir):
    """walk -> device parse -> device Huffman gives the samples the host parser gives (which the oracle pins)"""
    L = mlib.lib()
    for name, data in streams(golden_dir).items():
        w = mlib.walk_stream(data)
        if not w["regular"] or not w["n_frames"]:
            continue
        p = mlib.parse_stream(data)
        n, nch = w["n_frames"], w["channels"]
        d_img = ctx.to_device(np.frombuffer(data, dtype=np.uint8))
        d_refs, d_streams = ctx.to_device(w["refs"]), ctx.to_device(w["stream"])
        d_side, d_hdr, d_blob = ctx.alloc(n * 104), ctx.alloc(n * 8), ctx.alloc(w["blob_len"] + 16)
        d_st = ctx.to_device(np.zeros(4, dtype=np.int32))
        d_is, d_si, d_hst = ctx.alloc(n * 2304 * 2), ctx.alloc(n * 4 * 72), ctx.alloc(16)
        mlib.check(L.mp3s_parse_frames_dev(ctx.handle, d_img, 0, d_refs, d_streams, n, 0, d_side, d_hdr, d_blob, None, d_st))
        mlib.check(L.mp3s_huffman_decode_dev(ctx.handle, d_blob, d_side, n, nch, w["max_part2_3_length"], d_is, d_si, d_hst))
        got = ctx.download(d_is, np.int16, (n, 2, 2, 576))
        st = int(ctx.download(d_hst, np.int32, (1,))[0])
        if st == 0:
            assert np.array_equal(got[:, :, :nch], p["is"][:, :, :nch]), name
        for q in (d_img, d_refs, d_streams, d_side, d_hdr, d_blob, d_st, d_is, d_si, d_hst):
            ctx.free(q)
