"""The margin of the int16 decode's guard, MEASURED on the device (round-5 verdict, item 2).

k_dec_stream forms x = pcm * 32767 from sums that are not the reference's (DCT-IV halves, mirrored matrixing, forward-accumulated window:
DESIGN 2) and truncates it as decoder/MP3_Parser.py:91 truncates the reference's; an analytical bound eps_t (csrc/mp3s_tables.cpp) says which
samples may differ and must be recomputed in the reference's order (decoder/Frame.py:65-154).  tests/test_fast_synth.py checks the fix-up path;
this file checks that the bound is SUFFICIENT and by how much: the probe mp3s_debug_guard_margin makes the kernel leave x and eps_t per sample,
and r = |x_fast - x_exact| / eps_t (x_exact = the exact kernels' float64 PCM -- bit-identical to the reference, tests/test_gpu_parity.py --
times 32767) must stay <= 0.5 on every stream kind, an adversarial one included (tests/golden/gen_guard_adversarial.py: single lines whose
amplitude puts samples within ~1e-9 of non-zero integers).  The per-stream histogram of r goes to gpurun_out/guard_margin.json
(copied to profiles/r06_guard_margin.json)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGES = [0.0, 1e-4, 1e-3, 1e-2, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 1.0, np.inf]
RESULTS = {}


def margin(ctx, mlib, decode, n_samples):
    """decode(fmt) -> pcm; returns the statistics of r over the call's samples"""
    if not (ctx.get_option("fused_decode") and ctx.get_option("fast_imdct")):
        pytest.skip("the probe is an instantiation of the stream kernel, which this context's settings switch off (tools/option_sweep.sh)")
    exact = np.asarray(decode(mlib.MP3S_PCM_F64), dtype=np.float64).reshape(-1)
    assert exact.size == n_samples
    ctx.synth_mode(1.0)
    piped = ctx.get_option("file_pipeline")
    ctx.set_option("file_pipeline", 0)                  # one batch per call: the probe's index is the sample's index in the call's PCM
    try:
        with ctx.guard_margin(n_samples) as probe:
            i16 = np.asarray(decode(mlib.MP3S_PCM_I16)).reshape(-1)
            x, eps = probe.read()
    finally:
        ctx.set_option("file_pipeline", piped)
    fixed = ctx.synth_mode(1.0)                         # samples the guard sent through the exact order
    want = exact * 32767.0
    # the probe's index is the sample's index: what the kernel truncated is what it stored (the fix-up may have replaced flagged ones)
    near = np.abs(np.abs(x) - np.maximum(np.abs(np.rint(x)), 1.0)) <= eps
    with np.errstate(invalid="ignore"):
        trunc = np.where(np.abs(x) < 2147483648.0, x, 0.0).astype(np.int64).astype(np.int32).astype(np.int16)   # (the kernel's own conversion is checked elsewhere)
    ok = (trunc == i16) | near | ~np.isfinite(eps)
    assert ok.all(), int((~ok).sum())
    err = np.abs(x - want)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(np.isfinite(eps), np.where(eps > 0, err / eps, np.where(err == 0, 0.0, np.inf)), 0.0)
    worst = int(np.argmax(r))
    hist, _ = np.histogram(r, bins=EDGES)
    return {"samples": int(r.size), "max_r": float(r[worst]), "worst": {"index": worst, "x_fast": float(x[worst]), "x_exact": float(want[worst]),
                                                                      "eps_t": float(eps[worst])},
            "r_histogram": {"edges": [e if np.isfinite(e) else "inf" for e in EDGES], "counts": hist.tolist()},
            "guarded_granules_inf_eps": int((~np.isfinite(eps)).sum()), "samples_within_eps_of_an_integer": int(near.sum()),
            "recomputed_by_fixup": int(fixed), "max_abs_error": float(err[np.isfinite(eps)].max() if np.isfinite(eps).any() else 0.0)}


def record(name, res):
    RESULTS[name] = res
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(RESULTS, open(os.path.join(out, "guard_margin.json"), "w"), indent=1)
    except OSError:
        pass
    assert res["max_r"] <= 0.5, (name, res["max_r"], res["worst"])


def stream_case(ctx, mlib, name, data):
    n = ctx.decode_stream(data, mlib.MP3S_PCM_I16)["pcm"].size
    record(name, margin(ctx, mlib, lambda fmt: ctx.decode_stream(data, fmt)["pcm"], n))


def test_margin_on_reference_file_and_corpus(ctx, mlib, golden_dir):
    stream_case(ctx, mlib, "tests/test.mp3", open(os.path.join(golden_dir, "test.mp3"), "rb").read())
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    worst = None
    for k in sorted(g.files):
        if not k.endswith("__mp3"):
            continue
        data = g[k].tobytes()
        n = ctx.decode_stream(data, mlib.MP3S_PCM_I16)["pcm"].size
        res = margin(ctx, mlib, lambda fmt: ctx.decode_stream(data, fmt)["pcm"], n)
        res["stream"] = k[:-5]
        if worst is None or res["max_r"] > worst["max_r"]:
            tot = (worst or {}).get("corpus_samples", 0)
            worst = dict(res, corpus_samples=tot + res["samples"])
        else:
            worst["corpus_samples"] += res["samples"]
    record("g7 decode corpus (worst stream)", worst)


def test_margin_on_bench_stream_noise_and_escapes(ctx, mlib):
    from synth_pcm import synth_pcm
    import frame_synth
    rng = np.random.default_rng(8)
    stream_case(ctx, mlib, "10000 frames 44.1 kHz stereo 128 kbit/s (bench stream)",
                bytes(ctx.encode_pcm(synth_pcm(10000, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"]))
    loud = rng.integers(-32768, 32767, size=(400 * 1152, 2)).astype(np.int16)
    stream_case(ctx, mlib, "full-scale noise 48 kHz 320 kbit/s", bytes(ctx.encode_pcm(loud, 48000, 320, None)["mp3"]))
    stream_case(ctx, mlib, "mono, escape values up to 8206", frame_synth.make_stream(32, 90, mode=3, max_lin=8191))
    stream_case(ctx, mlib, "short / mixed blocks, MS stereo", frame_synth.make_stream(31, 150, block_types=(0, 1, 2, 3), allow_mixed=True, mode=1, mode_ext=2))


def test_margin_on_adversarial_batch(ctx, mlib, golden_dir):
    """single lines whose amplitude was searched so that one sample per active granule and channel lies within ~1e-9 of a non-zero integer"""
    a = np.load(os.path.join(golden_dir, "g9_guard_adversarial.npz"))
    n_gran = int(a["granule"].max()) + 3
    n = (n_gran + 1) // 2
    isv = np.zeros((n, 2, 2, 576), dtype=np.int16)
    si = np.zeros((n, 2, 2), dtype=mlib.GRANULE_SI_DTYPE)
    si["global_gain"] = 210
    for g, ch, line, val, gg in zip(a["granule"], a["channel"], a["line"], a["value"], a["global_gain"]):
        isv[g // 2, g % 2, ch, line] = val
        si["global_gain"][g // 2, g % 2, ch] = gg
    hdr = np.zeros(n, dtype=mlib.FRAME_HDR_DTYPE)
    hdr["nch"] = 2
    res = margin(ctx, mlib, lambda fmt: ctx.decode_transform(isv, si, hdr, 2, 0, fmt), n * 2304)
    # the premise, on the exact kernels' own values: the searched samples are where the generator put them
    exact = np.asarray(ctx.decode_transform(isv, si, hdr, 2, 0, mlib.MP3S_PCM_F64), dtype=np.float64).reshape(-1, 2) * 32767.0
    at = exact[a["granule"].astype(np.int64) * 576 + a["sample"], a["channel"]]
    dist = np.abs(at - np.rint(at))
    assert (np.abs(np.rint(at)) >= 1).all() and np.allclose(at, a["x_predicted"], rtol=0, atol=1e-7)
    res["searched_samples"] = int(dist.size)
    res["searched_within_1e-9"] = int((dist <= 1e-9).sum())
    res["searched_within_1e-8"] = int((dist <= 1e-8).sum())
    res["searched_worst_distance"] = float(dist.max())
    assert res["searched_within_1e-8"] >= 0.9 * dist.size and res["searched_within_1e-9"] >= 20, res
    # every one of them is inside its guard's width, i.e. decided by the exact order, and the int16 PCM is the exact kernels'
    ctx.synth_mode(1.0)
    i16 = np.asarray(ctx.decode_transform(isv, si, hdr, 2, 0, mlib.MP3S_PCM_I16)).reshape(-1, 2)
    ctx.synth_mode(0.0)
    want = np.asarray(ctx.decode_transform(isv, si, hdr, 2, 0, mlib.MP3S_PCM_I16)).reshape(-1, 2)
    ctx.synth_mode(1.0)
    assert np.array_equal(i16, want)
    record("adversarial: single lines, samples within 1e-9 of non-zero integers", res)


def test_probe_refuses_a_decode_that_would_not_fill_it(ctx, mlib):
    """with the stream kernel switched off the int16 decode takes the two kernels, which carry no probe: refused while the probe is set, not left unfilled"""
    import frame_synth
    data = frame_synth.make_stream(5, 40)
    n = ctx.decode_stream(data, mlib.MP3S_PCM_I16)["pcm"].size
    fused, piped = ctx.get_option("fused_decode"), ctx.get_option("file_pipeline")
    ctx.set_option("file_pipeline", 0)                          # (as every use of the probe: one batch per call)
    try:
        with ctx.guard_margin(n):
            ctx.set_option("fused_decode", 0)
            with pytest.raises(mlib.Mp3sError, match="guard probe"):
                ctx.decode_stream(data, mlib.MP3S_PCM_I16)
            ctx.decode_stream(data, mlib.MP3S_PCM_F64)          # (float formats never fill it and pass)
    finally:
        ctx.set_option("fused_decode", fused)
        ctx.set_option("file_pipeline", piped)
    ctx.decode_stream(data, mlib.MP3S_PCM_I16)                  # the probe is gone: nothing is refused
