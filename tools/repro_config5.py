#!/usr/bin/env python3
"""Why a decode of config 5's joint_ms_short_48k_192 mix takes 1.65 or 2.46 ms on the same code (round-5 verdict, item 3): the history of bench.py's
one-file context replayed (10 000- and 100 000-frame files hidden in and decoded, the mixes in front of the joint one decoded to int16 and
float32), then the joint mix decoded N times: every call's time and the address of the page-locked block its PCM came down into.
usage: python tools/repro_config5.py [calls=16] [history=1]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mp3stego import _lib
from synth_pcm import synth_pcm
import frame_synth
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 16
history = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.Context(0)
out = {"history": history}
if history:
    mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=7), 44100, 128, None)["mp3"])
    fs = _lib.parse_stream(mp3)["frame_size"]
    big = mp3[:int(fs[:9999].sum())] * 10
    for _ in range(3):
        r = ctx.hide_message(mp3, "hello"); del r
    for _ in range(3):
        r = ctx.hide_message(big, "hello"); del r
    for _ in range(3):
        r = ctx.decode_file(big); del r
    for _ in range(3):
        r = ctx.decode_stream(mp3, _lib.MP3S_PCM_I16); del r
def timed(name, kw, n):
    seed = kw.pop("seed")
    data = frame_synth.make_stream(seed, 250, **kw) * 40
    row = {}
    for fmt, key in ((_lib.MP3S_PCM_I16, "int16"), (_lib.MP3S_PCM_F32, "float32")):
        ts, blocks = [], []
        for _ in range(n):
            t0 = time.perf_counter(); r = ctx.decode_stream(data, fmt); dt = time.perf_counter() - t0
            blocks.append(hex(r["pcm"].ctypes.data >> 20)); del r
            ts.append(round(dt * 1e3, 3))
        row[key + "_ms"] = ts
        row[key + "_block_mb_address"] = blocks
    out[name] = row
    out[name]["pinned_pooled_mb"] = round(ctx.host_share()["pinned_pooled_bytes"] / 2 ** 20, 1)
if history:
    timed("switching_reservoir_44k_128", dict(seed=103, block_types=(0, 1, 2, 3), use_reservoir=True), 8)
    timed("mixed_blocks_44k_128", dict(seed=104, block_types=(0, 2), allow_mixed=True, use_reservoir=True), 8)
mode = os.environ.get("REPRO_MODE", "A")
out["mode"] = mode
if mode == "C":      # a pause with the device idle, then a mix that has been decoded before
    time.sleep(0.05)
    timed("switching_again_after_50_ms_idle", dict(seed=103, block_types=(0, 1, 2, 3), use_reservoir=True), calls)
if mode == "B":      # the joint mix through the exact kernels first (no stream kernel, no fix-up), then through the fast ones
    ctx.synth_mode(0.0)
    timed("joint_exact_kernels", dict(seed=105, sr_idx=1, bitrate_idx=11, mode=1, mode_ext=2, block_types=(0, 2), use_reservoir=True), calls)
    ctx.synth_mode(1.0)
if mode in ("F", "G"):      # the pipe re-made for the larger frames by a SHORT file of them; F: then 100 ms of nothing; G: straight on
    short = frame_synth.make_stream(105, 250, sr_idx=1, bitrate_idx=11, mode=1, mode_ext=2, block_types=(0, 2), use_reservoir=True) * 16    # 4 000 frames: two chunks
    t0 = time.perf_counter(); r = ctx.decode_stream(short, _lib.MP3S_PCM_I16); del r
    out["short_file_first_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    if mode == "F":
        time.sleep(0.1)
if mode == "D":
    ctx.set_option("file_pipeline", 0)
if mode == "E":      # a stereo (not joint) stream of the same rate and blocks in front of it
    timed("stereo_short_48k_192", dict(seed=105, sr_idx=1, bitrate_idx=11, block_types=(0, 2), use_reservoir=True), calls)
timed("joint_ms_short_48k_192", dict(seed=105, sr_idx=1, bitrate_idx=11, mode=1, mode_ext=2, block_types=(0, 2), use_reservoir=True), calls)
print(json.dumps(out))
