"""The stream index (mp3s_index_stream) and scans of frame ranges (mp3s_scan_range): a block scanned from the resume point
in front of it equals the corresponding part of the whole-file scan -- side records, main data (bit reservoir included),
headers, frame sizes, stego bits (incl. the stale table_select[2] of window-switching granules).  Host only."""
import os

import numpy as np
import pytest


def _frames(sc):
    return [bytes(sc["blob"][int(s["md_off"]):int(s["md_off"]) + int(s["md_len"])]) for s in sc["side"]]


def _check(mlib, mp3, ranges):
    full = mlib.scan_stream(mp3)
    ix = mlib.StreamIndex(mp3)
    n = full["n_frames"]
    assert ix.n_frames == n and ix.gpu_ok == full["gpu_ok"] and ix.dup_last_frame == full["dup_last_frame"]
    assert (ix.channels, ix.sampling_rate, ix.bit_rate) == (full["channels"], full["sampling_rate"], full["bit_rate"])
    ff = _frames(full)
    for a, cnt in ranges:
        r = ix.scan_range(a, cnt)
        k = r["n_frames"]
        assert k == max(0, min(cnt, n - a)), (a, cnt)
        assert _frames(r) == ff[a:a + k], (a, cnt)
        s1, s2 = r["side"].copy(), full["side"][a:a + k].copy()
        s1["md_off"] = 0
        s2["md_off"] = 0
        assert np.array_equal(s1, s2), (a, cnt)
        assert np.array_equal(r["hdr"], full["hdr"][a:a + k]) and np.array_equal(r["frame_size"], full["frame_size"][a:a + k])
        assert np.all(r["side"]["md_off"] % 4 == 0)
    # consecutive ranges give the stream's stego bits back
    step = max(1, n // 7)
    bits = [ix.scan_range(a, step)["bits"] for a in range(0, max(n, 1), step)]
    assert np.array_equal(np.concatenate(bits) if bits else np.zeros(0, np.uint8), full["bits"])
    ix.close()


def test_ranges_of_the_reference_fixture(mlib, golden_dir):
    mp3 = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    _check(mlib, mp3, [(0, 36), (0, 1), (35, 1), (35, 9), (17, 5), (36, 3), (100, 1)])


def test_ranges_across_resume_points_with_reservoir_and_window_switching(mlib):
    """streams longer than the distance between resume points (256 frames): reservoir reaching back over a resume point,
    window-switching granules whose third table index is inherited from frames in front of it"""
    import frame_synth
    rng = np.random.default_rng(3)
    for seed, kw in ((1, dict(block_types=(0, 1, 2, 3), use_reservoir=True)), (2, dict(block_types=(0, 2), mode=1, mode_ext=2, use_reservoir=True, crc=True)),
                     (3, dict(block_types=(0,), mode=3)), (4, dict(block_types=(0, 1, 3), sr_idx=1, bitrate_idx=5, use_reservoir=True))):
        mp3 = frame_synth.make_stream(seed, 600, **kw)
        n = mlib.scan_stream(mp3)["n_frames"]
        assert n >= 590
        ranges = [(0, n), (255, 2), (256, 1), (257, 300), (511, 3), (512, 88), (n - 1, 1), (n - 5, 50)]
        ranges += [(int(a), int(c)) for a, c in zip(rng.integers(0, n, 12), rng.integers(1, 300, 12))]
        _check(mlib, mp3, ranges)


def test_ranges_of_damaged_streams(mlib, golden_dir):
    """a bad header in the middle (the reference repeats the last frame, D12), a truncated file, an ID3 tag in front"""
    mp3 = bytearray(open(os.path.join(golden_dir, "test.mp3"), "rb").read())
    cut = bytes(mp3[:20000])
    _check(mlib, cut, [(0, 50), (10, 5), (18, 4)])
    bad = bytearray(mp3)
    bad[1045 * 20] = 0x00                         # frame 20 has no sync
    _check(mlib, bytes(bad), [(0, 50), (19, 1), (15, 10)])
    tag = b"ID3\x03\x00\x00\x00\x00\x00\x0a" + b"\x00" * 10
    _check(mlib, tag + bytes(mp3), [(0, 36), (30, 10)])
    with pytest.raises(mlib.Mp3sError):
        mlib.StreamIndex(b"\xff\xfb\x90")


def test_ranges_of_mutants(mlib, golden_dir):
    """bit flips in headers, side info and main data (false syncs, frame sizes that change, reservoir pointers into nowhere):
    whatever the whole-file scan makes of a mutant, the index and the range scans make the same of it"""
    import frame_synth
    rng = np.random.default_rng(77)
    bases = [open(os.path.join(golden_dir, "test.mp3"), "rb").read(),
             frame_synth.make_stream(5, 300, block_types=(0, 1, 2, 3), use_reservoir=True, allow_mixed=True),
             frame_synth.make_stream(6, 280, mode=3, use_reservoir=True, crc=True)]
    checked = rejected = 0
    for base in bases:
        for _ in range(60):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            m = bytes(b)
            try:
                full = mlib.scan_stream(m)
            except mlib.Mp3sError as e:
                with pytest.raises(mlib.Mp3sError) as e2:
                    mlib.StreamIndex(m)
                assert e2.value.code == e.code
                rejected += 1
                continue
            n = full["n_frames"]
            ranges = [(0, n + 3), (max(0, n - 2), 5)] + [(int(a), int(c)) for a, c in zip(rng.integers(0, max(n, 1), 4), rng.integers(1, 280, 4))]
            _check(mlib, m, ranges)
            checked += 1
    assert checked > 100
