"""The bench line's contract, checked on the latest line committed under profiles/ (bench.py itself needs the GPU):
metric and unit are BASELINE.json's, every field the driver reads is there, roofline and cpu_baseline are complete and
consistent with each other."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8000.0


def test_latest_bench_line_has_the_contract_fields():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))[-1]
    line = json.loads(open(latest).read().strip().splitlines()[-1])
    assert line["metric"] == base["metric"]
    if base.get("unit"):
        assert line["unit"] in base["unit"] or base["unit"] in line["unit"]
    for key, typ in (("value", (int, float)), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", (int, float)),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["scaling"] == "weak" and "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    # achieved = algorithmic bytes per launch / the dominant kernel's duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_per_launch"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # whole-job value = frames of all ranks / step time
    assert abs(line["value"] - line["config"]["frames_per_gpu"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.01
    assert line["parity_checked"] is True
    # round 2 and later: the binding roofline, the copy kernel, and the host-fed regions, steps-averaged
    if "e2e_steady" in line:
        a = line["roofline_alu"]
        assert a["bound"] == "valu_issue" and 0 < a["frac"] < 1 and abs(a["frac"] - a["achieved"] / a["peak"]) < 1e-3
        assert a["kernel"] == r["kernel"] and a["valu_wave_instructions_per_launch"] > 0
        assert r["copy_kernel_gbs"] and 1000 < r["copy_kernel_gbs"] < HBM_PEAK
        e = line["e2e_steady"]
        assert e["steps"] >= 200 and e["frames_per_s"] > 0
        assert abs(e["frames_per_s"] - line["config"]["frames_per_gpu"] * line["n_gpus"] / (e["ms_per_batch"] * 1e-3)) / e["frames_per_s"] < 0.01
        for k in ("h2d_kernels_d2h", "bytes_to_bytes_one_at_a_time", "decode_steady"):
            assert line["regions"][k]["batches"] >= 10 and line["regions"][k]["ms_per_batch"] > 0, k
        assert line["decode_only"]["steps"] >= 10 and line["config"]["chain_verdict_units_to_redo"] == 0
    # round 4: where the traffic and instruction counts come from, the sustained region, what a context's first call costs
    if "sustained" in line and line["sustained"]:
        u = line["sustained"]
        assert u["seconds"] >= 4.5 and u["batches"] > 1000 and 0.9 < u["last_over_first"] < 1.1
        assert abs(u["frames_per_s"] - line["config"]["frames_per_gpu"] * line["n_gpus"] * u["batches"] / u["seconds"]) / u["frames_per_s"] < 0.01
        assert "device" in u and "sclk_mhz" in u["device"]
        assert r["traffic_source"] and line["roofline_alu"]["pmc_source"]
        fc = line["regions"]["single_file_10k"]["first_call"]
        assert fc["rehearsals"] <= 12 and fc["ms"] > fc["rehearsal_ms"] >= 0

