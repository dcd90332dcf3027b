"""Blocks and chunks of a stream scanned on their own (stream index, include/mp3s.h): indexed block calls equal the
whole-file ones, a file streamed through the device chunk by chunk gives the bytes of the one-call functions (SURVEY 8f n4,
second half)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_indexed_blocks_equal_whole_file_blocks(ctx, mlib, golden_dir):
    from synth_pcm import synth_pcm
    import frame_synth
    streams = [bytes(ctx.encode_pcm(synth_pcm(700, seed=31), 44100, 128, None)["mp3"]),
               frame_synth.make_stream(7, 560, block_types=(0, 1, 2, 3), use_reservoir=True),           # reservoir, all block types
               frame_synth.make_stream(8, 300, mode=3, use_reservoir=True),                              # mono
               open(os.path.join(golden_dir, "test.mp3"), "rb").read()]
    for si, mp3 in enumerate(streams):
        ix = mlib.StreamIndex(mp3)
        n = ix.n_frames
        whole = ctx.decode_stream(mp3, mlib.MP3S_PCM_F64)
        for first, cnt in ((0, 5), (1, 1), (255, 3), (256, 40), (n - 3, 10), (n // 2, n)):
            if first < 0 or first >= n:
                continue
            a = ctx.decode_block(mp3, first, cnt, mlib.MP3S_PCM_F64)
            b = ctx.decode_block(mp3, first, cnt, mlib.MP3S_PCM_F64, index=ix)
            assert a["n_frames"] == b["n_frames"] and a["pcm"].tobytes() == b["pcm"].tobytes(), (si, first, cnt)
        # the stream in chunks, each scanned on its own
        for chunk in (100, 257):
            got = np.concatenate(list(ctx.decode_chunks(mp3, chunk, mlib.MP3S_PCM_F64)))
            assert got.tobytes() == whole["pcm"].tobytes(), (si, chunk)


def test_chunked_hide_equals_one_call(ctx, mlib, orc):
    from synth_pcm import synth_pcm
    pcm = synth_pcm(900, seed=41)
    pcm[:40 * 1152] = 0                                      # the stream starts in silence (inherited state crosses chunk boundaries)
    pcm[300 * 1152:330 * 1152] = 0
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    cut = mp3[:-777]                                         # a truncated last frame: the decoder repeats the last PCM frame
    for data in (mp3, cut):
        for msg in ("short", "x" * 300, None):
            whole = ctx.clear_file(data) if msg is None else ctx.hide_message(data, msg)
            for chunk in (64, 257, 512, 5000):
                got = ctx.hide_message_chunked(data, msg, chunk)
                assert bytes(got["data"]) == bytes(whole["data"]), (len(data), msg and len(msg), chunk)
                assert got["too_long"] == whole["too_long"] and got["hide_offset"] == whole["hide_offset"] and got["n_frames"] == whole["n_frames"]
    # against the oracle directly
    d = orc.decode(mp3)
    o = orc.encode(orc.pcm_to_i16(d["pcm"]), 44100, 128, np.array(mlib.message_frame("short")))
    assert bytes(ctx.hide_message_chunked(mp3, "short", 100)["data"]) == o["mp3"]
    # a stream the host parser has to take (mixed blocks): the chunks fall back to whole-file scans, same bytes
    import frame_synth
    mixed = frame_synth.make_stream(9, 40, block_types=(0, 2), allow_mixed=True)
    assert not mlib.StreamIndex(mixed).gpu_ok
    want = ctx.decode_stream(mixed, mlib.MP3S_PCM_I16)["pcm"]
    assert np.array_equal(np.concatenate(list(ctx.decode_chunks(mixed, 16))), want)
