"""N > 1 path of bench.py on CPU: world size 2 over gloo.  The data path has no collective (frames shard
embarrassingly); what is exercised is the rendezvous, the barrier and the max-over-ranks reduction, plus the
frame-sharding arithmetic with its 1-frame halo."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(%r, "tests"))
    from shard import shard_frames
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n = 1001
    first, count, halo = shard_frames(n, rank, world)
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([count], dtype=torch.int64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"max": float(t[0]), "total": int(tot[0])}))
    print("SHARD", rank, first, count, halo)
    dist.destroy_process_group()
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_arithmetic():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shard import shard_frames
    for n in (1, 7, 8, 1000, 1001, 10 ** 6):
        for world in (1, 2, 4, 8):
            covered = 0
            for r in range(world):
                first, count, halo = shard_frames(n, r, world)
                assert first == covered and count >= 0
                assert halo == (1 if (first > 0 and count > 0) else 0)
                covered += count
            assert covered == n


def test_two_ranks_gloo(tmp_path):
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
    import json
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][0]
    res = json.loads(line)
    assert res["max"] == 1.5 and res["total"] == 1001
    shards = sorted(tuple(map(int, l.split()[1:])) for o, _ in outs for l in o.splitlines() if l.startswith("SHARD"))
    assert shards == [(0, 0, 501, 0), (1, 501, 500, 1)]
