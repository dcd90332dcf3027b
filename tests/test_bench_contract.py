"""The bench line's contract, checked on the latest line committed under profiles/ (bench.py itself needs the GPU):
metric and unit are BASELINE.json's, every field the driver reads is there, roofline and cpu_baseline are complete and
consistent with each other."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8000.0


def test_latest_bench_line_has_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    if not lines:
        import pytest
        pytest.skip("no bench line under profiles/ (a checkout without it)")
    latest = lines[-1]
    check_line(json.loads(open(latest).read().strip().splitlines()[-1]))


def check_line(line, full=True):
    """`full`: a default run (every region, 200+ batches of the steady state); otherwise a shortened run of the same script"""
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
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
    if not full:
        assert line["parity_checked"] is True and line["e2e_steady"]["frames_per_s"] > 0 and line["regions"]["single_file_10k"]["ms_per_batch"] > 0
        return
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


import subprocess
import sys

import pytest


@pytest.mark.gpu
def test_a_live_run_keeps_the_contract():
    """bench.py itself, shortened (2 000 frames, 5 steps, 1 s of `sustained`, 1 s of CPU baseline), on the test box's GPU: one json line
    on stdout, every field the driver reads, roofline and cpu_baseline consistent, traffic and instruction counts measured in the
    run (the committed line above is checked for the full-size numbers; this catches a bench.py that no longer runs)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--frames", "2000", "--no-config5",
                        "--no-single-file-100k", "--e2e-batches", "40", "--sustained-seconds", "1", "--cpu-seconds", "1", "--cpu-seconds-all", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    check_line(line, full=False)
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["config"]["frames_per_gpu"] == 2000
    assert line["sustained"]["batches"] > 100 and "sclk_mhz" in line["sustained"]["device"]
    # (where rocprofv3 is installed the counters are this run's own)
    if os.path.exists("/opt/rocm/bin/rocprofv3"):
        assert line["roofline"]["traffic_source"].startswith("live") and line["roofline"]["traffic"] > 0, line["roofline"]
        assert line["roofline_alu"]["pmc_source"].startswith("live")

