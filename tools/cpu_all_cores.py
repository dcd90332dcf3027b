"""All-cores leg of bench.py's cpu_baseline: the oracle's decode -> int16 -> encode on every host core, one
forked worker per core, each looping over its own 400-frame cut of the stream.  Runs as a child process of bench.py
(which has a GPU context and must not fork); prints one JSON object.

    python tools/cpu_all_cores.py <stream.mp3> <hide_bits.npy> <seconds>
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
K = 400


def _work(job):
    import oracle_lib as O
    chunk, hide, t_end = job
    frames = 0
    t_start = time.time()
    while frames == 0 or time.time() < t_end:
        od = O.decode(chunk)
        oe = O.encode(O.pcm_to_i16(od["pcm"]), int(od["sampling_rate"]), int(od["bit_rate"]) // 1000, hide)
        assert oe["rc"] == 0
        frames += od["n_frames"]
    return frames / (time.time() - t_start)


def main():
    data = open(sys.argv[1], "rb").read()
    hide = np.load(sys.argv[2])
    seconds = float(sys.argv[3])
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # frame boundaries: walk the headers the way the encoder laid them out (no reservoir, no tags)
    offs = [0]
    pos = 0
    while pos + 4 <= len(data):
        pad = (data[pos + 2] >> 1) & 1
        br = (0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 0)[data[pos + 2] >> 4]
        sr = (44100, 48000, 32000, 0)[(data[pos + 2] >> 2) & 3]
        if data[pos] != 0xFF or not br or not sr:
            break
        pos += 144000 * br // sr + pad
        offs.append(pos)
    n = len(offs) - 1
    if n <= K:
        print(json.dumps({"error": "stream too short"}))
        return
    chunks = []
    for i in range(cores):
        a = (i * K) % (n - K + 1)
        chunks.append(data[offs[a]:offs[a + K]])
    t0 = time.time()
    t_end = t0 + seconds
    with mp.get_context("fork").Pool(cores) as pool:
        rate = sum(pool.map(_work, [(c, hide, t_end) for c in chunks], chunksize=1))   # sum of the workers' own rates
    dt = time.time() - t0
    print(json.dumps({"value": round(rate, 1), "unit": "frames/s", "cores": cores,
                      "sample": f"{cores} worker processes (one per schedulable CPU; a container CPU quota may allow fewer to run at "
                                f"once), each looping over its own {K}-frame cut of the stream, {dt:.1f} s"}))


if __name__ == "__main__":
    main()
