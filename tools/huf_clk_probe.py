"""Phase clocks of k_dec_huffman from a -DMP3S_HUF_CLOCKS=1 build (tools/ubench/build/libclk.so copied over the library): the
kernel leaves shader-clock deltas in sample pairs 280..285 of every row."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mp3-steganography-lib_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from mp3stego import _lib as m
from synth_pcm import synth_pcm
ctx = m.Context(); L = m.lib()
for n in [int(a) for a in sys.argv[1:]] or [8, 2048, 10000]:
    pcm = synth_pcm(n, seed=0x9E3779B97F4A7C15)
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
    w = m.walk_stream(mp3)
    d_img = ctx.to_device(np.frombuffer(mp3, dtype=np.uint8))
    d_refs, d_streams = ctx.to_device(w["refs"]), ctx.to_device(w["stream"])
    d_side, d_hdr, d_blob = ctx.alloc(n * 104), ctx.alloc(n * 8), ctx.alloc(w["blob_len"] + 16)
    d_st = ctx.to_device(np.zeros(4, dtype=np.int32))
    d_is, d_si, d_hst = ctx.alloc(n * 2304 * 2), ctx.alloc(n * 4 * 72), ctx.alloc(16 + 4 * n)
    m.check(L.mp3s_parse_frames_dev(ctx.handle, d_img, 0, d_refs, d_streams, n, 0, d_side, d_hdr, d_blob, None, d_st))
    for _ in range(5):
        m.check(L.mp3s_huffman_decode_dev(ctx.handle, d_blob, d_side, n, 2, w["max_part2_3_length"], d_is, d_si, d_hst))
    ctx.sync()
    rows = ctx.download(d_is, np.uint32, (n * 4, 288))[:, 280:286].astype(np.int64)
    side = ctx.download(d_side, m.FRAME_SIDE_DTYPE, (n,))
    bv = side["unit"]["big_values"].reshape(-1)
    print(n, "frames: clocks tables %d  stage+scalefactors %d  symbol loop %d  zero tail %d (medians); loop max %d; p_end median %d max %d; big_values median %d max %d; realtime span %.1f us"
          % (np.median(rows[:, 0]), np.median(rows[:, 1]), np.median(rows[:, 2]), np.median(rows[:, 3]), rows[:, 2].max(), np.median(rows[:, 4]), rows[:, 4].max(),
             np.median(bv), bv.max(), (rows[:, 5].max() - rows[:, 5].min()) / 100.0))
