"""tools/launch_ranks.py: `bench.py --gpus N` / `tools/config4.py --procs N` start their own ranks (one process
per GPU, no collective on the data path).  The argument / environment / exit-code logic on CPU with a stand-in child; the
real thing runs in tests/test_sharded.py (-m gpu)."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import launch_ranks  # noqa: E402

CHILD = textwrap.dedent("""
    import json, os, sys, time
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(r) and os.environ["LOCAL_WORLD_SIZE"] == str(w)
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    mode = sys.argv[1]
    print("stderr of rank", r, file=sys.stderr)
    if mode == "gloo":
        import torch, torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=r, world_size=w)
        t = torch.tensor([float(r + 1)]); dist.all_reduce(t)
        dist.barrier(); dist.destroy_process_group()
        if r == 0:
            print(json.dumps({"n_gpus": w, "sum": float(t[0])}))
        else:
            print("noise from rank", r)             # must not reach the job's stdout
    elif mode == "fail1":
        if r == 1:
            sys.exit(7)
        time.sleep(600)                             # the others would wait in a barrier for ever
    elif mode == "ok":
        if r == 0:
            print(json.dumps({"n_gpus": w}))
""")


def test_plan_rules():
    ok, _ = launch_ranks.plan(8, {}, 8)
    assert ok
    ok, msg = launch_ranks.plan(8, {}, 1)
    assert not ok and "MP3STEGO_DEVICE" in msg
    ok, _ = launch_ranks.plan(8, {"MP3STEGO_DEVICE": "0"}, 1)          # launch-path check on a 1-GPU box
    assert ok
    ok, _ = launch_ranks.plan(2, {}, -1)                               # device count unknown and nothing pinned
    assert not ok
    ok, _ = launch_ranks.plan(0, {}, 8)
    assert not ok


def test_child_env():
    e = launch_ranks.child_env({"A": "b", "RANK": "9"}, 3, 4, 1234)
    assert e["A"] == "b" and e["RANK"] == "3" and e["LOCAL_RANK"] == "3" and e["WORLD_SIZE"] == "4"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "1234" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_too_many_ranks_is_a_clear_error(tmp_path):
    (tmp_path / "c.py").write_text(CHILD)
    out, err = io.StringIO(), io.StringIO()
    rc = launch_ranks.launch([sys.executable, str(tmp_path / "c.py"), "ok"], 8, env={k: v for k, v in os.environ.items() if k != "MP3STEGO_DEVICE"},
                             n_devices=1, out=out, err=err)
    assert rc == launch_ranks.E_USAGE and out.getvalue() == "" and "--gpus 8" in err.getvalue()


def _run(tmp_path, mode, n, **kw):
    (tmp_path / "c.py").write_text(CHILD)
    # (real file objects: the children inherit stderr)
    with open(tmp_path / "out.txt", "w+") as out, open(tmp_path / "err.txt", "w+") as err:
        rc = launch_ranks.launch([sys.executable, str(tmp_path / "c.py"), mode], n, env=dict(os.environ, MP3STEGO_DEVICE="0"),
                                 out=out, err=err, **kw)
        out.seek(0); err.seek(0)
        return rc, out.read(), err.read()


def test_ranks_rendezvous_and_only_rank0_speaks(tmp_path):
    rc, out, err = _run(tmp_path, "gloo", 3, timeout=300)
    assert rc == 0, err[-2000:]
    # (the gloo transport announces its connections on the stdout of the process that initialises it; bench.py moves that to stderr)
    lines = [l for l in out.splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 3, "sum": 6.0} and "noise" not in out
    assert "noise from rank 1" in err and "stderr of rank 2" in err


def test_a_dead_rank_ends_the_job_with_its_code(tmp_path):
    rc, out, err = _run(tmp_path, "fail1", 2, grace=1.0, timeout=120)
    assert rc == 7 and "rank 1 exited with 7" in err


def test_timeout(tmp_path):
    rc, out, err = _run(tmp_path, "fail1", 1, timeout=1.0)
    assert rc == launch_ranks.E_TIMEOUT


def test_bench_py_refuses_more_ranks_than_devices():
    """the parent's path in bench.py itself: no GPU here, so --gpus 2 without MP3STEGO_DEVICE must end with the usage code and no
    json line (the old bench.py ignored --gpus and printed n_gpus 1)"""
    env = {k: v for k, v in os.environ.items() if k not in ("MP3STEGO_DEVICE", "WORLD_SIZE", "RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == launch_ranks.E_USAGE and r.stdout.strip() == "" and "--gpus 2" in r.stderr
