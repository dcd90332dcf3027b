"""Spectra (MDCT lines, int32 [n][576]) that PCM through the encoder's filter bank does not produce: lone lines, empty regions below the last
big value, everything at the quantiser's first thresholds.  What the rate loop's shortcuts (csrc/k_rate.hpp rl_precheck, the bounds in rl_body)
are checked on, against the reference's loop body restated in oracle/ (MP3_Encoder.py:958-996, 1064-1095, 1170-1264)."""
import numpy as np

KINDS = ("lone_line", "hf_few", "hf_noise", "lf_gap", "dense", "ones", "twos_far_apart", "count1_only")


def sparse_spectra(seed, n, base=32768):
    """n spectra, kinds in rotation.  `base`: |xr| that quantises to 1 at the step the caller aims at (32768 = step -60, the binary search's
    first probe: tests/test_tables.py pins rl_t1[67]); amplitudes are drawn around small multiples of it so that a probe sees 0, 1, 2, 3 ..."""
    rng = np.random.default_rng(seed)
    xr = np.zeros((n, 576), dtype=np.int64)

    def amp(size, lo=0.3, hi=40.0):
        return (base * np.exp(rng.uniform(np.log(lo), np.log(hi), size))).astype(np.int64)

    def signs(size):
        return rng.integers(0, 2, size) * 2 - 1
    for i in range(n):
        k = KINDS[i % len(KINDS)]
        if k == "lone_line":                       # one line, mostly far up: big_values reaches it, every region below is empty
            p = int(rng.integers(0, 576)) if rng.random() < 0.3 else int(rng.integers(400, 576))
            xr[i, p] = amp(1, 1.0, 200.0)[0] * signs(1)[0]
        elif k == "hf_few":                        # a handful of lines above line 300
            m = int(rng.integers(1, 12))
            p = rng.choice(np.arange(300, 576), m, replace=False)
            xr[i, p] = amp(m, 0.8, 30.0) * signs(m)
        elif k == "hf_noise":                      # nothing below a cut, low-level noise above
            cut = int(rng.integers(100, 560))
            m = 576 - cut
            xr[i, cut:] = amp(m, 0.2, 6.0) * signs(m) * (rng.random(m) < rng.uniform(0.05, 1.0))
        elif k == "lf_gap":                        # a low band, a gap of zeros, a high band
            a, b = sorted(rng.integers(2, 570, 2))
            xr[i, :a] = amp(a, 0.3, 20.0) * signs(a)
            xr[i, b:] = amp(576 - b, 0.3, 8.0) * signs(576 - b)
        elif k == "dense":                         # every line, one level per spectrum
            lvl = np.exp(rng.uniform(np.log(0.5), np.log(3000.0)))
            xr[i] = (base * lvl * rng.random(576)).astype(np.int64) * signs(576)
        elif k == "ones":                          # everything quantises to 0 or 1 at the aimed step
            xr[i] = amp(576, 0.5, 1.9) * signs(576) * (rng.random(576) < rng.uniform(0.02, 1.0))
        elif k == "twos_far_apart":                # a few values >= 2 between long runs of zeros and ones
            xr[i] = amp(576, 0.5, 1.9) * signs(576) * (rng.random(576) < rng.uniform(0.0, 0.3))
            m = int(rng.integers(1, 5))
            p = rng.choice(576, m, replace=False)
            xr[i, p] = amp(m, 2.5, 12.0) * signs(m)
        else:                                      # count1_only: ones in the lowest lines only
            top = int(rng.integers(1, 200))
            xr[i, :top] = amp(top, 0.9, 1.9) * signs(top) * (rng.random(top) < 0.7)
    return np.clip(xr, -(2 ** 31 - 1), 2 ** 31 - 1).astype(np.int32)
