"""The rate loop's threshold pre-check (csrc/k_rate.hpp rl_precheck) decides a probe of the binary search from a LOWER BOUND of its bits.
Here the bound is restated in numpy from the library's own threshold tables and held against the reference's loop body (oracle/:
MP3_Encoder.py:973-990 with quantize :373-415, calc_run_len :266-291, count1_bit_count :171-211, __new_choose_table :1170-1264) on spectra
with empty regions below the last big value -- where a bound that charges every big-value pair a bit is NOT one (round-5 advisor finding:
table 0 costs nothing, :1182-1184, :228-229).  CPU only; the device's decisions on the same spectra: tests/test_gpu_parity.py."""
import numpy as np
import pytest

from spectra import sparse_spectra


def precheck_model(xr, t1, t2):
    """(big_values, count1, lower bound as the kernel forms it, the round-5 form) per spectrum"""
    xa = np.abs(xr.astype(np.int64))
    nz, big = xa >= t1, xa >= t2
    pnz, pbig = nz[:, 0::2] | nz[:, 1::2], big[:, 0::2] | big[:, 1::2]
    idx = np.arange(288)[None, :]
    p0 = np.where(pnz.any(1), (pnz * (idx + 1)).max(1) - 1, -1)
    p1 = np.where(pbig.any(1), (pbig * (idx + 1)).max(1) - 1, -1)
    count1 = (p0 - p1) >> 1
    bv = p0 + 1 - 2 * count1
    nnz = nz.sum(1)
    below = (pnz & (idx < bv[:, None])).sum(1)
    return bv, count1, nnz + below + count1, nnz + bv + count1


@pytest.mark.parametrize("step", [-60, -90, -30, -45, -100, -8])
def test_precheck_bound_is_a_lower_bound(mlib, orc, step):
    t = mlib.debug_tables()
    i = step + 127
    t1, t2, t8 = int(t["rl_t1"][i]), int(t["rl_t2"][i]), int(t["rl_t8"][i])
    if t1 == 0xffffffff:
        pytest.skip("nothing quantises to 1 at this step")
    xr = sparse_spectra(0xB0 + i, 6000, base=max(t1, 1))
    bits, bv, c1 = orc.probe_bits(44100, step, xr)
    mbv, mc1, lb, lb_r5 = precheck_model(xr, t1, t2)
    ran = (bits >= 0) & (bits < 100000)                                # the loop body ran (quantize did not refuse, not silent)
    assert ran.sum() > 4000
    # quantize's refusal exactly where the threshold says (:409-410)
    xmax = np.abs(xr.astype(np.int64)).max(1)
    assert np.array_equal(bits == 100000, (xmax >= t8) & (bits >= 0))
    # the run lengths the pre-check leaves are the reference's
    assert np.array_equal(mbv[ran], bv[ran]) and np.array_equal(mc1[ran], c1[ran])
    # ... and its sum never exceeds the probe's bits
    assert (lb[ran] <= bits[ran]).all(), int((lb[ran] > bits[ran]).sum())
    if step == -60:
        # the premise: these spectra do reach the case in which a bit per big-value pair is too much
        assert (lb_r5[ran] > bits[ran]).sum() > 50
