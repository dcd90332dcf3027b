#!/usr/bin/env python3
"""N ranks on ONE host, the host's side only: N processes, each walking its own 10 000-frame file over
and over -- the host stage of a pipe job since round 3 (frame walk; side info and main data are taken apart on the device)
-- for a few seconds, no GPU work.  Prints, for N = 1, 2, 4, 8 (and what the quota allows): frames/s per process and in
all, against the 12.2 M frames/s a device consumes, plus what the library would pick per rank (CPUs allowed, scan threads).

usage: python tools/host_scale_probe.py [seconds=2] > profiles/r03_host_scale.json        (GPU box or any host)"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mp3-steganography-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
DEVICE_RATE = 12.2e6          # frames/s one MI355X takes bytes -> bytes (bench.py e2e_steady)


def cgroup_quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


def worker(args):
    path, seconds, mode = args
    from mp3stego import _lib
    data = open(path, "rb").read()
    if mode == "walk":
        r, n = _lib.walk_rate(data, seconds)
        return r
    t0 = time.perf_counter()
    frames = 0
    while time.perf_counter() - t0 < seconds:
        frames += _lib.scan_stream(data)["n_frames"]
    return frames / (time.perf_counter() - t0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    import tempfile
    import numpy as np
    import oracle_lib as O
    from synth_pcm import synth_pcm
    # a 10 000-frame 128 kbit/s file without a GPU: the oracle encodes 250 frames, laid end to end (every frame stands alone)
    e = O.encode(synth_pcm(251, seed=5), 44100, 128, None)
    from mp3stego import _lib
    fs = _lib.parse_stream(e["mp3"])["frame_size"]
    one = e["mp3"][:int(fs[:250].sum())]
    data = one * 40
    td = tempfile.mkdtemp()
    path = os.path.join(td, "s.mp3")
    open(path, "wb").write(data)
    out = {"host_cpus_visible": os.cpu_count(), "cpus_allowed": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": cgroup_quota(),
           "frames_per_file": 10000, "device_rate_frames_per_s": DEVICE_RATE, "seconds": seconds, "runs": []}
    for mode in ("walk",):
        for n in (1, 2, 4, 8, 16):
            with mp.get_context("fork").Pool(n) as pool:
                rates = pool.map(worker, [(path, seconds, mode)] * n)
            total = sum(rates)
            out["runs"].append({"host_stage": mode, "processes": n, "frames_per_s_each_min": round(min(rates)), "frames_per_s_each_max": round(max(rates)),
                                "frames_per_s_all": round(total), "devices_this_feeds": round(total / DEVICE_RATE, 2),
                                "ms_of_one_core_per_10k_frames": round(1e4 / (total / n) * 1e3, 4)})
    walk8 = next(r for r in out["runs"] if r["host_stage"] == "walk" and r["processes"] == 8)
    out["eight_ranks_fed"] = walk8["frames_per_s_all"] >= 8 * DEVICE_RATE
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
