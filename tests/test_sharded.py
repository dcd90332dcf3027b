"""One stream over several ranks (mp3stego/sharded.py, mp3s_decode_block, mp3s_encode_block): blocks of
frames with a one-frame halo / lead and the 17-integer carry between them must reproduce, bit for bit, what one GPU
makes of the whole stream -- which the other tests pin to the oracle."""
import json
import time
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _play(world, fn):
    """the ranks of a LocalComm one after the other; the last call returns the gathered result"""
    from mp3stego import sharded
    comm = sharded.LocalComm(world)
    out = None
    for r in range(world):
        comm.rank = r
        out = fn(comm)
    return out


@pytest.mark.gpu
def test_decode_blocks_equal_whole_stream(ctx, mlib, golden_dir):
    from mp3stego import sharded
    g = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    streams = [open(os.path.join(golden_dir, "test.mp3"), "rb").read()]
    streams += [g[n + "__mp3"].tobytes() for n in names]             # mono, MS, reservoir, mixed blocks, CRC, ID3 ...
    streams.append(streams[0][:-700])                                # ends inside a frame: the last PCM frame is repeated
    for data in streams:
        whole = ctx.decode_stream(data, mlib.MP3S_PCM_F64)
        n = whole["n_frames"]
        for world in (1, 2, 3, 7):
            got = _play(world, lambda comm: sharded.decode_sharded(ctx, data, comm, mlib.MP3S_PCM_F64))
            assert got["pcm"].shape == whole["pcm"].shape and got["pcm"].tobytes() == whole["pcm"].tobytes(), (n, world)
            assert np.array_equal(got["bits"], whole["bits"])
        # single blocks anywhere, clipped at the end of the stream
        for first, cnt in ((0, 1), (n - 1, 5), (n // 2, 2)):
            if first < 0:
                continue
            b = ctx.decode_block(data, first, cnt, mlib.MP3S_PCM_F64)
            k = min(cnt, n - first)
            tail = whole["pcm"][first * 1152:] if first + cnt >= n else whole["pcm"][first * 1152:(first + k) * 1152]
            assert b["n_frames"] == k and b["pcm"].tobytes() == tail.tobytes(), (first, cnt)
        with pytest.raises(mlib.Mp3sError):
            ctx.decode_block(data, n, 1)                             # starts behind the last frame


@pytest.mark.gpu
def test_reencode_blocks_equal_whole_stream(ctx, mlib, orc):
    """boundaries in sound, in silence (inherited address / quantizer state crosses them), inside the message"""
    from mp3stego import sharded
    from synth_pcm import synth_pcm
    rng = np.random.default_rng(5)
    pcm = synth_pcm(180, seed=21)
    pcm[55 * 1152:65 * 1152] = 0                                     # silent around frame 60 = boundary of world 3
    pcm[88 * 1152:93 * 1152, 0] = 0                                  # one channel silent around frame 90 = world 2
    cases = [(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"], [None, "short", "m" * 150, "".join(chr(int(c)) for c in rng.integers(32, 127, size=700))]),
             (ctx.encode_pcm(synth_pcm(50, rate=48000, seed=3), 48000, 192, None)["mp3"], ["x" * 40]),
             (ctx.encode_pcm(synth_pcm(31, rate=32000, seed=4), 32000, 64, None)["mp3"][:-300], [None, "tail"]),
             # 2 whole frames + a cut one: the decoder repeats the last PCM frame (D12); with 3 and 6 ranks one block ends
             # on the stream's last frame and the next one is the repeated frame alone
             (ctx.encode_pcm(synth_pcm(3, seed=9), 44100, 128, None)["mp3"][:-200], [None, "x"])]
    for mp3, msgs in cases:
        for msg in msgs:
            whole = ctx.clear_file(mp3) if msg is None else ctx.hide_message(mp3, msg)
            for world in (1, 2, 3, 6):
                got = _play(world, lambda comm: sharded.reencode_sharded(ctx, mp3, msg, comm))
                assert got["data"] == whole["data"], (len(mp3), msg and len(msg), world)
                assert got["too_long"] == whole["too_long"] and got["hide_offset"] == whole["hide_offset"]
    # and against the oracle directly for one of them
    mp3, msg = cases[0][0], "short"
    d = orc.decode(mp3)
    o = orc.encode(orc.pcm_to_i16(d["pcm"]), 44100, 128, np.array(mlib.message_frame(msg)))
    assert _play(3, lambda comm: sharded.reencode_sharded(ctx, mp3, msg, comm))["data"] == o["mp3"]


@pytest.mark.gpu
def test_encode_block_carry_contract(ctx, mlib):
    """the C entry point on its own: a block that starts in silence depends on the carry and says so; one that starts
    in sound does not; block bytes concatenate to the stream"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(40, seed=8)
    pcm[18 * 1152:24 * 1152] = 0
    whole = ctx.encode_pcm(pcm, 44100, 128, None)
    a = ctx.encode_block(pcm[:20 * 1152], 0, 0, False, 44100, 128)
    assert not a["carry_used"] and a["carry_out"][1:].any()
    zero = np.zeros(17, dtype=np.int64)
    b_guess = ctx.encode_block(pcm[19 * 1152:], 1, 20, True, 44100, 128, None, zero)
    assert b_guess["carry_used"]                                     # frame 20 is silent: it inherits
    b = ctx.encode_block(pcm[19 * 1152:], 1, 20, True, 44100, 128, None, a["carry_out"])
    assert a["mp3"] + b["mp3"] == whole["mp3"] and a["mp3"] + b_guess["mp3"] != whole["mp3"]
    c1 = ctx.encode_block(pcm[:10 * 1152], 0, 0, False, 44100, 128)
    c2 = ctx.encode_block(pcm[9 * 1152:], 1, 10, True, 44100, 128, None, zero)
    assert not c2["carry_used"] and c1["mp3"] + c2["mp3"] == whole["mp3"]
    for bad in (dict(lead=1, first=0, carry=None), dict(lead=0, first=5, carry=None), dict(lead=3, first=5, carry=zero)):
        with pytest.raises(mlib.Mp3sError):
            ctx.encode_block(pcm[:10 * 1152], bad["lead"], bad["first"], True, 44100, 128, None, bad["carry"])


def _shrink_part2_3(mp3, frame_sizes, frame, to):
    """part2_3_length of granule 0 / channel 0 of `frame` set to `to` bits: its big values then run past the end of the
    granule, which the device Huffman kernel reports and the host parser -- one bit cursor per frame, as the reference --
    decodes.  (stereo, no CRC: the 12-bit field starts 20 bits into the side info)"""
    at = int(np.sum(frame_sizes[:frame])) + 4
    b = bytearray(mp3)
    word = int.from_bytes(b[at:at + 4], "big")
    assert (word & 0xfff) > to
    word = (word & ~0xfff) | to
    b[at:at + 4] = word.to_bytes(4, "big")
    return bytes(b)


@pytest.mark.gpu
def test_blocks_of_a_stream_that_needs_the_host_parser(ctx, mlib, orc):
    """a frame the device Huffman decoder hands back to the host parser, sitting in the middle of a block: the block
    paths cut the host parser's frames to the same window (found by the soak: the fallback used to see the whole file)"""
    from mp3stego import sharded
    from synth_pcm import synth_pcm
    mp3 = ctx.encode_pcm(synth_pcm(30, seed=44), 44100, 128, None)["mp3"]
    sizes = mlib.scan_stream(mp3)["frame_size"]
    for frame in (7, 19, 29):
        bad = _shrink_part2_3(mp3, sizes, frame, 40)
        sc = mlib.scan_stream(bad)                                   # the device decoder does flag this stream
        d_blob, d_side = ctx.to_device(sc["blob"]), ctx.to_device(sc["side"])
        d_is, d_si, d_st = ctx.alloc(30 * 2304 * 2), ctx.alloc(30 * 4 * 72), ctx.alloc(4)
        mlib.check(mlib.lib().mp3s_huffman_decode_dev(ctx.handle, d_blob, d_side, 30, 2, sc["max_part2_3_length"], d_is, d_si, d_st))
        assert int(ctx.download(d_st, np.int32, (1,))[0]) != 0
        for q in (d_blob, d_side, d_is, d_si, d_st):
            ctx.free(q)
        whole = ctx.decode_stream(bad, mlib.MP3S_PCM_F64)
        o = orc.decode(bad)
        assert o["rc"] == 0 and whole["pcm"].tobytes() == o["pcm"].tobytes()
        # in a batch only the stream that holds the flagged frame goes back to the host parser; all come out right
        trio = ctx.decode_streams([mp3, bad, mp3[:418 * 9]], mlib.MP3S_PCM_F64)
        assert trio[1]["pcm"].tobytes() == o["pcm"].tobytes()
        assert trio[0]["pcm"].tobytes() == orc.decode(mp3)["pcm"].tobytes()
        assert trio[2]["pcm"].tobytes() == orc.decode(mp3[:418 * 9])["pcm"].tobytes()
        for world in (2, 3, 5):
            got = _play(world, lambda comm: sharded.decode_sharded(ctx, bad, comm, mlib.MP3S_PCM_F64))
            assert got["pcm"].tobytes() == whole["pcm"].tobytes(), (frame, world)
            again = ctx.hide_message(bad, "through the fallback")
            got = _play(world, lambda comm: sharded.reencode_sharded(ctx, bad, "through the fallback", comm))
            assert got["data"] == again["data"], (frame, world)


@pytest.mark.gpu
def test_reencode_block_entry_point(ctx, mlib):
    """mp3s_reencode_block on its own: blocks partition the PCM frames, ranks beyond the last frame get nothing, the
    blocks run on their real carries concatenate to the single-GPU file, argument errors are reported"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(23, seed=12)
    pcm[9 * 1152:14 * 1152] = 0
    mp3 = ctx.encode_pcm(pcm, 44100, 128, None)["mp3"][:-100]       # 22 whole frames + the repeated one = 23 PCM frames
    whole = ctx.hide_message(mp3, "block by block")
    for world in (1, 2, 5, 23, 30):
        carry, data, covered = None, b"", 0
        for rank in range(world):
            b = ctx.reencode_block(mp3, "block by block", rank, world, carry)
            assert b["total_frames"] == 23 and b["first_frame"] == covered
            covered += b["n_frames"]
            if b["n_frames"] == 0:
                assert rank >= 23 and b["mp3"] == b""
                carry = np.zeros(17, dtype=np.int64) if carry is None else carry
                continue
            data += b["mp3"]
            carry = b["carry_out"]
            assert b["is_last"] == (covered == 23)
        assert covered == 23 and data == whole["data"], world
        assert b["hide_offset"] == whole["hide_offset"] or b["n_frames"] == 0
    for rank, world, carry in ((0, 0, None), (2, 2, None), (-1, 2, None), (1, 2, None), (0, 2, np.zeros(17, dtype=np.int64))):
        with pytest.raises(mlib.Mp3sError):
            ctx.reencode_block(mp3, None, rank, world, carry)


WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, os.path.join(%r, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np
    import torch.distributed as dist
    from mp3stego import _lib, sharded
    dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    ctx = _lib.Context(0)                      # both ranks share the one GPU of the test box
    mp3 = open(sys.argv[1], "rb").read()
    comm = sharded.TorchComm()
    r = sharded.reencode_sharded(ctx, mp3, "two ranks, one stream", comm)
    d = sharded.decode_sharded(ctx, mp3, comm, _lib.MP3S_PCM_I16)
    if comm.rank == 0:
        print(json.dumps({"mp3": hashlib.sha256(r["data"]).hexdigest(), "too_long": bool(r["too_long"]),
                          "pcm": hashlib.sha256(d["pcm"].tobytes()).hexdigest()}))
    else:
        assert r is None and d is None
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, ROOT)


@pytest.mark.gpu
def test_two_processes_one_stream(ctx, mlib, tmp_path):
    """the real thing in small: two processes, a gloo group, the carry over dist.send/recv, blocks gathered on rank 0"""
    import hashlib
    from synth_pcm import synth_pcm
    pcm = synth_pcm(101, seed=33)
    pcm[48 * 1152:53 * 1152] = 0                                     # the boundary (frame 51) lies in silence
    mp3 = ctx.encode_pcm(pcm, 44100, 128, None)["mp3"]
    (tmp_path / "in.mp3").write_bytes(mp3)
    (tmp_path / "worker.py").write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(tmp_path / "worker.py"), str(tmp_path / "in.mp3")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    whole = ctx.hide_message(mp3, "two ranks, one stream")
    assert res["mp3"] == hashlib.sha256(whole["data"]).hexdigest() and res["too_long"] == whole["too_long"]
    assert res["pcm"] == hashlib.sha256(ctx.decode_stream(mp3, mlib.MP3S_PCM_I16)["pcm"].tobytes()).hexdigest()


def _json_line(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the shape of the driver's N=1 command): the parent starts two rank
    processes (tools/launch_ranks.py), both pinned to the test box's one GPU, and relays rank 0's one json line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MP3STEGO_DEVICE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--no-config5", "--no-cpu-baseline",
                        "--no-single-file-100k", "--e2e-batches", "40", "--frames", "2000"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["parity_checked"] is True and out["scaling"] == "weak"
    assert len(out["ranks_on_this_host"]) == 2 and sorted(h["rank"] for h in out["ranks_on_this_host"]) == [0, 1]
    assert out["config"]["parallelism"].startswith("frames sharded over 2 GPU")


@pytest.mark.gpu
def test_five_ranks_side_by_side_on_one_device():
    """The launch shape of the driver's multi-GPU run before the hardware sees it: five rank processes (a test box lets one user open its card from six processes at once, and
    the test runner is one of them; the driver's node takes eight ranks) with a context, a pipe and a rehearsal each, all on device 0.  Every rank reports what it
    holds of the host: its share of the page-locked pool (4 GB / ranks, at least 1 GB), the CPUs it may use and those on the GPU's node."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MP3STEGO_DEVICE"] = "0"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--steps", "3", "--no-config5", "--no-cpu-baseline",
                        "--no-single-file-100k", "--no-short-files", "--e2e-batches", "20", "--sustained-seconds", "0", "--frames", "2000"],
                       env=env, capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["n_gpus"] == 5 and out["parity_checked"] is True
    hosts = out["ranks_on_this_host"]
    assert sorted(h["rank"] for h in hosts) == list(range(5))
    for h in hosts:
        assert h["local_world_size"] == 5 and h["cpus_allowed"] >= 1
        assert h["pinned_pool_cap_bytes"] == max(1 << 30, (4 << 30) // 5) and h["pinned_pooled_bytes"] <= h["pinned_pool_cap_bytes"]
    assert sum(h["pinned_pooled_bytes"] for h in hosts) <= 5 << 30
    assert wall < 240, wall
    tool = os.path.join(ROOT, "tools", "config4.py")
    one = subprocess.run([sys.executable, tool, "5", "300", "5"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    five = subprocess.run([sys.executable, tool, "5", "300", "5", "--procs", "5"], env=env, capture_output=True, text=True, timeout=900)
    assert five.returncode == 0, five.stderr[-3000:]
    a, b = _json_line(one.stdout), _json_line(five.stdout)
    assert a["ok"] and b["ok"] and b["procs"] == 5 and a["rank_crc32"] == b["rank_crc32"]


@pytest.mark.gpu
def test_config4_as_processes_equals_config4_in_one_process():
    """BASELINE configs[3] in small: 5 streams x 400 frames as 2 blocks (the boundary halves stream 2).  One process playing
    both ranks and two processes (own context + pipe each, the carry over gloo) must print the same CRC per rank, and both
    must agree with the single call on every stream and with the oracle on the sample"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MP3STEGO_DEVICE"] = "0"
    tool = os.path.join(ROOT, "tools", "config4.py")
    one = subprocess.run([sys.executable, tool, "5", "400", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a = _json_line(one.stdout)
    two = subprocess.run([sys.executable, tool, "5", "400", "2", "--procs", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _json_line(two.stdout)
    assert a["ok"] and b["ok"] and b["procs"] == 2 and len(b["rank_ms"]) == 2
    assert a["rank_crc32"] == b["rank_crc32"]
    assert b["pieces"] == 6 and b["pieces_equal_to_single_call"] == 6 and b["halves_fit"]
