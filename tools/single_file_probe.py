#!/usr/bin/env python3
"""One file per call through the overlapped stages: ms per file for a 10 000-frame and a 100 000-frame file (hide / clear /
decode), against the stages one after the other (file_pipeline = 0), and the pipe's steady state per scan-thread count.
usage: python tools/single_file_probe.py [frames_small frames_large]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from mp3stego import _lib                      # noqa: E402
from synth_pcm import synth_pcm                # noqa: E402


def timed(f, n):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
        del r
    return (time.perf_counter() - t0) / n * 1e3


def main():
    n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_large = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    ctx = _lib.Context(0)
    mp3 = bytes(ctx.encode_pcm(synth_pcm(n_small, seed=7), 44100, 128, None)["mp3"])
    fs = _lib.parse_stream(mp3)["frame_size"]
    whole = mp3[:int(fs[:n_small - 1].sum())]                      # complete frames only: copies can follow one another
    big = whole * (n_large // (n_small - 1))
    n_big = (n_small - 1) * (n_large // (n_small - 1))
    out = {"device": ctx.device_name()}
    msg = "The quick brown fox jumps over the lazy dog, again & again, 0123"
    for name, data, n, reps in (("small", mp3, n_small, 30), ("large", big, n_big, 6)):
        row = {"frames": n, "bytes": len(data)}
        for label, opt in (("pipelined", 1), ("one_after_the_other", 0)):
            ctx.set_option("file_pipeline", opt)
            s0 = ctx.run_stats()
            row[label] = {"hide_ms": round(timed(lambda: ctx.hide_message(data, msg), reps), 4),
                          "clear_ms": round(timed(lambda: ctx.clear_file(data), reps), 4),
                          "decode_file_ms": round(timed(lambda: ctx.decode_file(data), max(3, reps // 3)), 4)}
            s1 = ctx.run_stats()
            row[label]["hide_frames_per_s"] = round(n / (row[label]["hide_ms"] * 1e-3))
            row[label]["run_stats_delta"] = {k: s1[k] - s0[k] for k in s1}
        ctx.set_option("file_pipeline", 1)
        for chunk in (2048, 4096, 8192, 12288):
            ctx.set_option("chunk_frames", chunk)
            row["chunk_%d" % chunk] = round(timed(lambda: ctx.hide_message(data, msg), reps), 4)
        ctx.set_option("chunk_frames", 0)
        out[name] = row
    # the pipe: jobs of one small file, steady state, per scan-thread count
    pctx = _lib.Context(0)
    for th in (1, 2, 3):
        pipe = _lib.Pipe(pctx, depth=4, max_job_bytes=len(mp3) + (1 << 16), scan_threads=th)
        nb, sub, got = 300, 0, 0
        t0 = None
        while got < nb + 20:
            while sub < nb + 20 and pipe.submit([mp3], [msg]) is not None:
                sub += 1
            _t, res = pipe.collect()
            del res
            got += 1
            if got == 20:
                t0 = time.perf_counter(); st0 = pipe.stats()
        dt = time.perf_counter() - t0
        st = pipe.stats()
        pipe.close()
        out["pipe_threads_%d" % th] = {"ms_per_batch": round(dt / nb * 1e3, 4), "frames_per_s": round(n_small * nb / dt),
                                       "host_walk_ms_per_batch": round((st["scan_ms"] - st0["scan_ms"]) / nb, 4),
                                       "host_issue_ms_per_batch": round((st["issue_ms"] - st0["issue_ms"]) / nb, 4), "fast": st["fast"], "slow": st["slow"]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
