#!/usr/bin/env python3
"""Soak of the rate loop on spectra PCM does not produce (tests/spectra.py: lone lines, empty regions below the last big value, everything at the
quantiser's first thresholds, dense noise at any level), aimed at random probes of the binary search, random (sampling rate, bit rate): every unit's
step, run lengths, regions, tables, bits and quantised lines of mp3s_rate_loop_dev against the reference's loop restated in the oracle
(orc_enc_rate_units: MP3_Encoder.py:766-813).  The shortcuts under test: rl_precheck and the lower / upper bounds of rl_body (csrc/k_rate.hpp).
usage (GPU box): python tools/soak_rate_spectra.py [seconds=300]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mp3stego import _lib as mlib
import oracle_lib as orc
from spectra import sparse_spectra
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
ctx = mlib.Context(0)
L = mlib.lib()
t = mlib.debug_tables()
rng = np.random.default_rng(606 + int(os.environ.get("SOAK_SEED", "0")))
t0 = t_say = time.time()
units_done = bad = batches = step_range = 0
FIELDS = (("quantizer_step", "quantizerStepSize"), ("big_values", "big_values"), ("count1", "count1"), ("part2_3_length", "part2_3_length"),
          ("region0_count", "region0_count"), ("region1_count", "region1_count"), ("count1table_select", "count1table_select"), ("table_select", "table_select"))
while time.time() - t0 < seconds:
    rate = int(rng.choice([32000, 44100, 48000]))
    kbps = int(rng.choice([32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]))
    n = 512
    units = 4 * n
    parts = []
    for k in range(4):
        st = int(rng.integers(-119, -1))                                  # the probe the spectra's levels are aimed at
        base = max(int(t["rl_t1"][st + 127]), 1)
        if base == 0xffffffff:
            base = 1 << 30
        parts.append(sparse_spectra(int(rng.integers(1 << 31)), units // 4, base=min(base, 1 << 27)))
    xr = np.ascontiguousarray(np.concatenate(parts)[rng.permutation(units)])
    rf, _ = mlib.rate_frames(rate, kbps, 2, n)
    # every other batch with a message: each unit sees it from its start (cursor 0), so its up to three tables go through the swap (MP3_Encoder.py:1257-1263)
    hide = rng.integers(0, 2, 3).astype(np.uint8) if batches % 2 else None
    want = orc.rate_units(rate, np.repeat(rf["max_bits"], 4), xr, hide)
    if hide is not None:
        rf["hide_end"] = len(hide)
    d_mdct, d_rf = ctx.to_device(xr), ctx.to_device(rf)
    d_hide = ctx.to_device(hide) if hide is not None else None
    d_cur = ctx.to_device(np.zeros(units, dtype=np.int32)) if hide is not None else None
    d_ix, d_out, d_en = ctx.alloc(units * 576 * 2), ctx.alloc(units * 72), ctx.alloc(units * 22 * 4)
    d_state = ctx.to_device(np.zeros((units, 4), dtype=np.int32))
    mlib.check(L.mp3s_rate_loop_dev(ctx.handle, d_mdct, d_rf, n, d_hide, 0 if hide is None else len(hide), d_cur, d_state, None, 0, d_ix, d_out, d_en))
    ctx.sync()
    out = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
    ix = ctx.download(d_ix, np.int16, (units, 576)).astype(np.int32)
    for p in (d_mdct, d_rf, d_ix, d_out, d_en, d_state, d_hide, d_cur):
        if p is not None:
            ctx.free(p)
    gi = want["gi"]
    ok = want["rc"] == 0
    wrong = ((out["flags"] & mlib.RF_STEP_RANGE) != 0) != ~ok
    act = (np.abs(xr).max(1) > 0) & ok
    for a, b in FIELDS:
        wrong |= act & ~(out[a] == gi[b]).reshape(units, -1).all(1)
    for k, fld in enumerate(("address1", "address2", "address3")):
        wrong |= act & (out["address"][:, k] != gi[fld])
    wrong |= act & ~(np.abs(ix) == want["ix"]).all(1)
    wrong |= act & ~((ix == 0) | ((ix < 0) == (xr < 0))).all(1)
    nb = int(wrong.sum())
    if nb:
        print("MISMATCH", rate, kbps, None if hide is None else hide.tolist(), np.nonzero(wrong)[0][:8], flush=True)
    bad += nb; units_done += units; batches += 1; step_range += int((~ok).sum())
    if time.time() - t_say > 30:
        t_say = time.time()
        print("... %d units, %d mismatches" % (units_done, bad), file=sys.stderr, flush=True)
print(json.dumps({"units": units_done, "batches": batches, "mismatches": bad, "units_whose_step_left_the_table_in_both": step_range, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
