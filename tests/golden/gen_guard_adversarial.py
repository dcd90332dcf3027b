"""Generator of tests/golden/g9_guard_adversarial.npz: a decode batch built so that MANY samples of x = pcm * 32767 (what the reference
truncates to int16, decoder/MP3_Parser.py:91) lie within ~1e-9 of a non-zero integer -- the cases the fast int16 decode's guard exists for
(decoder/Frame.py:65-154 is the order of the sums whose truncation is at stake).

Every third granule of a stereo 44.1 kHz stream carries ONE non-zero line per channel (long blocks, no scalefactors): its PCM is the line's
amplitude A = |is|^(4/3) * 2^((global_gain - 210) / 4) (Frame.py:210-215) times a fixed response f(t) over the granule, the next one and 15
slots of the one after (IMDCT overlap, 16 slots of synthesis history) -- nothing of the neighbouring active granules reaches it.  f(t) is
taken from the oracle's stage functions (the reference's operation order), and for each active granule and channel the (line, is, global_gain)
among those tried whose A puts some sample closest to a non-zero integer is kept.  The file holds the choices and the predicted samples.
Run from the repo root: python tests/golden/gen_guard_adversarial.py   (about ten minutes of numpy; needs oracle/liborc.so, no GPU)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O

N_ACTIVE = 48          # active granules (every third granule)
LINES_TRIED = 6        # lines tried per active granule and channel
PEAK_LO, PEAK_HI = 1500.0, 30000.0


def response(L, line):
    prev, fifo = np.zeros(576), np.zeros(1024)
    out = []
    for k in range(3):
        smp = np.zeros(576)
        if k == 0:
            smp[line] = 1.0
        L.orc_alias_reduction(smp)
        L.orc_imdct(smp, 0, prev)
        L.orc_frequency_inversion(smp)
        L.orc_synth_filter_bank(smp, fifo)
        out.append(smp.copy())
    return np.concatenate(out)


def main():
    L = O.lib()
    rng = np.random.default_rng(0x6A2D)
    v_all = np.arange(1, 8192, dtype=np.float64)
    p43 = np.power(v_all, 4.0 / 3.0)
    rows = []
    for g in range(N_ACTIVE):
        for ch in range(2):
            best = None
            for line in rng.choice(576, LINES_TRIED, replace=False):
                f = response(L, int(line)) * 32767.0
                fmax = np.abs(f).max()
                keep = np.nonzero(np.abs(f) > 0.02 * fmax)[0]            # samples large enough to matter
                for gg in range(120, 256):
                    b = np.power(2.0, (gg - 210.0) / 4.0)
                    A = p43 * b
                    ok = (A * fmax >= PEAK_LO) & (A * fmax <= PEAK_HI)
                    if not ok.any():
                        continue
                    Ak = A[ok]
                    x = Ak[:, None] * f[keep][None, :]
                    r = np.rint(x)
                    d = np.where(np.abs(r) >= 1, np.abs(x - r), 1.0)
                    i, j = np.unravel_index(np.argmin(d), d.shape)
                    if best is None or d[i, j] < best[0]:
                        best = (float(d[i, j]), int(line), int(v_all[ok][i]), gg, int(keep[j]), float(x[i, j]))
            sign = -1 if rng.random() < 0.5 else 1
            rows.append((3 * g, ch, best[1], sign * best[2], best[3], best[4], sign * best[5], best[0]))
            print("granule %3d ch %d: line %3d is %5d gg %3d -> sample %4d x = %.12f (distance %.2e)" % rows[-1], flush=True)
    a = np.array(rows)
    np.savez_compressed(os.path.join(HERE, "g9_guard_adversarial.npz"),
                        granule=a[:, 0].astype(np.int32), channel=a[:, 1].astype(np.int32), line=a[:, 2].astype(np.int32),
                        value=a[:, 3].astype(np.int32), global_gain=a[:, 4].astype(np.int32), sample=a[:, 5].astype(np.int32),
                        x_predicted=a[:, 6].astype(np.float64), distance=a[:, 7].astype(np.float64))


if __name__ == "__main__":
    main()
