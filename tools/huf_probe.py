"""Duration of k_dec_huffman alone over the number of frames in the launch (the latency chain of one wave vs the fill of the
chip): `python tools/huf_probe.py [frames ...]`; MP3S_HUF_LANES picks the lane width."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mp3-steganography-lib_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from mp3stego import _lib as m
from synth_pcm import synth_pcm

ctx = m.Context()
L = m.lib()
sizes = [int(a) for a in sys.argv[1:]] or [8, 64, 256, 1024, 2048, 2560, 4096, 5000, 8192, 10000]
if os.environ.get("HUF_PROBE_STREAM"):      # a mix of tests/frame_synth.py instead of this encoder's own stream: long | short | switching | 320
    import frame_synth
    kw = {"long": dict(block_types=(0,)), "short": dict(block_types=(2,)), "switching": dict(block_types=(0, 1, 2, 3)), "320": dict(bitrate_idx=14, block_types=(0,))}[os.environ["HUF_PROBE_STREAM"]]
    mp3 = frame_synth.make_stream(101, 250, use_reservoir=True, **kw) * ((max(sizes) + 249) // 250)
else:
    pcm = synth_pcm(max(sizes), seed=0x9E3779B97F4A7C15)
    mp3 = bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"])
w = m.walk_stream(mp3)
res = {}
for n in sizes:
    refs = w["refs"][:n].copy()
    st = w["stream"].copy(); st["n_frames"] = n
    d_img = ctx.to_device(np.frombuffer(mp3, dtype=np.uint8))
    d_refs, d_streams = ctx.to_device(refs), ctx.to_device(st)
    blob_len = int(refs["md_off"][-1]) + int(refs["md_len"][-1]) + 8
    d_side, d_hdr, d_blob = ctx.alloc(n * 104), ctx.alloc(n * 8), ctx.alloc(blob_len + 16)
    d_st = ctx.to_device(np.zeros(4, dtype=np.int32))
    d_is, d_si, d_hst = ctx.alloc(n * 2304 * 2), ctx.alloc(n * 4 * 72), ctx.alloc(16 + 4 * n)
    m.check(L.mp3s_parse_frames_dev(ctx.handle, d_img, 0, d_refs, d_streams, n, 0, d_side, d_hdr, d_blob, None, d_st))
    ctx.sync()
    reps = 200
    for k in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            m.check(L.mp3s_huffman_decode_dev(ctx.handle, d_blob, d_side, n, 2, w["max_part2_3_length"], d_is, d_si, d_hst))
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
    res[n] = round(dt * 1e6, 1)
    for q in (d_img, d_refs, d_streams, d_side, d_hdr, d_blob, d_st, d_is, d_si, d_hst):
        ctx.free(q)
print("lanes", os.environ.get("MP3S_HUF_LANES", "auto"), "us per launch:", res)
