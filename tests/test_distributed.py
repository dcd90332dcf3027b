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


SHARD_WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%r, "mp3-steganography-lib_amd"))
    from mp3stego import sharded
    dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))

    class StubCtx:
        # stands in for the GPU context: a block that starts at frame 30 or later "inherits" (its bytes show the carry it got)
        calls = 0
        def reencode_block(self, mp3, message, rank, world, carry_in, index=None):
            from mp3stego import _lib
            StubCtx.calls += 1
            total = _lib.scan_stream(mp3)["n_frames"]
            first, n = sharded.shard_frames(total, rank, world)
            c = np.zeros(17, dtype=np.int64) if carry_in is None else np.array(carry_in, dtype=np.int64)
            used = first >= 30
            out = c.copy()
            out[0] = c[0] + 12 * n
            if not used:
                out[1:] = first + 1
            last = first + n == total
            text = "<%%d,%%d,%%s,%%d>" %% (first, n, ",".join(map(str, c)) if used else "-", int(last))
            return {"total_frames": total, "first_frame": first, "n_frames": n, "is_last": last, "carry_used": used,
                    "carry_out": out, "mp3": text.encode(), "kbps": 128, "sampling_rate": 44100, "too_long": False,
                    "hide_offset": int(out[0])}

    comm = sharded.TorchComm()
    mp3 = open(sys.argv[1], "rb").read()
    r = sharded.reencode_sharded(StubCtx(), mp3, None, comm)
    print("CALLS", comm.rank, StubCtx.calls)
    if comm.rank == 0:
        print(json.dumps({"data": r["data"].decode(), "hide_offset": r["hide_offset"]}))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_sharded_stream_plumbing_over_gloo(tmp_path):
    """mp3stego/sharded.py between real processes (gloo, world 3) with a stand-in for the GPU context: the carry travels
    rank to rank, a block that did not look at its carry is not run again, one that did is, rank 0 gets the blocks in
    order.  (The same code against the real library: tests/test_sharded.py on the GPU.)"""
    golden = os.path.join(ROOT, "tests", "golden", "g6_synth128.npz")
    mp3 = np.load(golden)["mp3"].tobytes()
    (tmp_path / "in.mp3").write_bytes(mp3)
    (tmp_path / "worker.py").write_text(SHARD_WORKER)
    port = free_port()
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(tmp_path / "worker.py"), str(tmp_path / "in.mp3")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    # 48 frames over 3 ranks: blocks of 16 at frames 0, 16, 32; the last one "inherits"
    c2 = [12 * 32] + [17] * 16                                       # rank 1 did not look at its carry: own chain, real count
    assert res["data"] == "<0,16,-,0><16,16,-,0><32,16,%s,1>" % ",".join(map(str, c2))
    assert res["hide_offset"] == 12 * 48
    calls = dict(tuple(map(int, l.split()[1:])) for o, _ in outs for l in o.splitlines() if l.startswith("CALLS"))
    assert calls == {0: 1, 1: 1, 2: 2}                               # only the block that inherits is run a second time


EIGHT_WORKER = textwrap.dedent("""
    import os, sys, json
    import torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "mp3-steganography-lib_amd"))
    import bench
    from mp3stego import _lib
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    steady = {"host_scan_ms_per_batch": 0.2 + rank, "host_issue_ms_per_batch": 0.1, "bytes_in_per_batch": 4e6, "bytes_out_per_batch": 4e6, "ms_per_batch": 1.0}
    hosts = bench.gather_rank_hosts(dist, world, bench.rank_host_record(rank, 1, steady, _lib.host_share))
    if rank == 0:
        print(json.dumps({"ranks_on_this_host": hosts}))
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def _run_ranks(script, args, world, extra_env=None, per_rank_cpus=None):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **(extra_env or {}))
        pre = None
        if per_rank_cpus:
            cpus = per_rank_cpus[rank]
            pre = lambda cpus=cpus: os.sched_setaffinity(0, cpus)             # noqa: E731  (what a launcher that pins its ranks would do)
        procs.append(subprocess.Popen([sys.executable, str(script)] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, preexec_fn=pre))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    return outs


def test_eight_ranks_carry_chain_over_gloo(tmp_path):
    """SURVEY 8e at the world size the driver launches (8), without eight GPUs: mp3stego/sharded.py between eight real processes over gloo
    with the stand-in context -- the carry (message cursor + four inherited chains, 136 bytes: encoder/MP3_Encoder.py:808-809, 1004-1006) passes
    through SEVEN hand-overs, the blocks that did not look at theirs run once, the three that did run again on the real one, rank 0 assembles
    the blocks in order."""
    import json
    golden = os.path.join(ROOT, "tests", "golden", "g6_synth128.npz")
    (tmp_path / "in.mp3").write_bytes(np.load(golden)["mp3"].tobytes())
    (tmp_path / "worker.py").write_text(SHARD_WORKER)
    outs = _run_ranks(tmp_path / "worker.py", [str(tmp_path / "in.mp3")], 8)
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    # 48 frames over 8 ranks: blocks of 6 at frames 0, 6, ..., 42; the stand-in "inherits" from frame 30 on.  Every block adds 72 to the cursor;
    # a block that did not inherit sets its own chains (first + 1), one that did passes on what it got
    want = "".join("<%d,6,-,0>" % (6 * r) for r in range(5))
    for r, cur in ((5, 360), (6, 432), (7, 504)):
        want += "<%d,6,%s,%d>" % (6 * r, ",".join(map(str, [cur] + [25] * 16)), int(r == 7))
    assert res["data"] == want
    assert res["hide_offset"] == 12 * 48
    calls = dict(tuple(map(int, l.split()[1:])) for o, _ in outs for l in o.splitlines() if l.startswith("CALLS"))
    assert calls == {0: 1, 1: 1, 2: 1, 3: 1, 4: 1, 5: 2, 6: 2, 7: 2}


def test_eight_ranks_share_the_host_and_rank0_assembles_the_line(tmp_path):
    """what eight ranks of one host are told (mp3s_ctx_host_share without a context: LOCAL_WORLD_SIZE = 8 -> the page-locked pool's cap is
    4 GB / 8 but at least 1 GB; the CPUs each may use, here one CPU per rank where the container has eight) and bench.py's own assembly of
    `ranks_on_this_host` on rank 0 (bench.rank_host_record / gather_rank_hosts), over gloo"""
    import json
    (tmp_path / "worker.py").write_text(EIGHT_WORKER)
    avail = sorted(os.sched_getaffinity(0))
    pins = [{avail[r % len(avail)]} for r in range(8)]
    outs = _run_ranks(tmp_path / "worker.py", [], 8, per_rank_cpus=pins)
    hosts = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])["ranks_on_this_host"]
    assert [h["rank"] for h in hosts] == list(range(8))
    for r, h in enumerate(hosts):
        assert h["local_world_size"] == 8 and h["pinned_pool_cap_bytes"] == 1 << 30 and h["pinned_pooled_bytes"] == 0
        assert h["cpus_allowed"] == 1 and h["gpu_node_cpus"] == 0
        assert h["host_walk_ms_per_batch"] == 0.2 + r and h["pcie_gb_s"] == 8.0
    if len(avail) >= 8:
        assert len({tuple(p) for p in pins}) == 8                       # (distinct CPUs per rank on this container)
