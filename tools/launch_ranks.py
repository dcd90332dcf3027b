"""One process per GPU, started from a plain `python script.py --gpus N` (no torch.distributed.run in front).

The parent makes NO GPU call and never exec()s: it starts N children of the same script (`subprocess.Popen`) with
RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's stdout (the ONE json line
of bench.py) and everybody's stderr, and returns the worst exit code.  A rank that dies takes the others with it after a
grace period (they would wait for it in a barrier for ever); only the exact PIDs started here are ever signalled.

Used by bench.py (`--gpus N` with WORLD_SIZE unset) and tools/config4.py (`--procs N`); the argument / exit-code logic
is covered on CPU by tests/test_launcher.py.
"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mp3-steganography-lib_amd")

E_USAGE = 2          # more ranks asked for than devices to put them on
E_TIMEOUT = 124


def visible_devices():
    """HIP devices a child of this process can open.  Asked in a short-lived child (mp3s_device_count touches the HIP
    runtime; this process must not).  -1 when the library cannot say (not built, no runtime)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mp3stego import _lib\n"
            "print(_lib.device_count())\n") % PKG
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else -1
    except Exception:                                    # noqa: BLE001
        return -1


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def plan(n, env, n_devices):
    """-> (ok, message).  N ranks need N devices unless MP3STEGO_DEVICE pins every rank to one (launch-path checks on a
    1-GPU box: the ranks then share it and the number is no scaling claim)."""
    if n < 1:
        return False, f"--gpus {n}: at least one rank"
    if env.get("MP3STEGO_DEVICE", "") != "":
        return True, f"{n} ranks pinned to device {env['MP3STEGO_DEVICE']} (MP3STEGO_DEVICE): they share it"
    if n_devices < 0:
        return False, "cannot count the HIP devices of this host (is the library built?)"
    if n > n_devices:
        return False, (f"--gpus {n} but this host shows {n_devices} HIP device(s); set MP3STEGO_DEVICE=<k> to put "
                       f"every rank on one device (a launch-path check, not a scaling run)")
    return True, f"{n} ranks on devices 0..{n - 1}"


def child_env(env, rank, n, port):
    e = dict(env)
    e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
             MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


def launch(argv, n, env=None, timeout=None, grace=30.0, n_devices=None, out=None, err=None):
    """start `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) N times, one rank each.
    -> worst exit code (E_USAGE without starting anything if the ranks do not fit the devices).
    timeout None = MP3STEGO_LAUNCH_TIMEOUT seconds (default 3600): a rank that hangs without exiting -- a rendezvous or a
    barrier that never completes -- ends the job with E_TIMEOUT instead of keeping the parent polling for ever.  A job whose
    ranks die within seconds without a line from rank 0 (the port taken between free_port() and the ranks' bind: the socket
    is closed before they start) is started once more, fresh children on a new port; nothing is ever re-executed."""
    env = dict(os.environ if env is None else env)
    if timeout is None:
        try:
            timeout = float(env.get("MP3STEGO_LAUNCH_TIMEOUT", "3600"))
        except ValueError:
            timeout = 3600.0
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    for attempt in range(2):
        t_start = time.time()
        code, said = _launch_once(argv, n, env, timeout, grace, n_devices, out, err)
        if code in (0, E_USAGE, E_TIMEOUT) or said or time.time() - t_start > 20.0 or attempt == 1:
            return code
        print(f"[launch_ranks] the ranks ended with {code} within {time.time() - t_start:.1f} s and rank 0 said nothing: once more on a new port",
              file=err, flush=True)
    return code


def _launch_once(argv, n, env, timeout, grace, n_devices, out, err):
    """-> (worst exit code, rank 0 printed something)"""
    if n_devices is None:
        n_devices = -1 if env.get("MP3STEGO_DEVICE", "") != "" else visible_devices()
    ok, msg = plan(n, env, n_devices)
    print(f"[launch_ranks] {msg}", file=err, flush=True)
    if not ok:
        return E_USAGE, False
    port = free_port()
    procs = []
    for r in range(n):
        # rank 0's stdout is the job's stdout; the other ranks' stdout joins stderr (one json line per job)
        procs.append(subprocess.Popen(argv, env=child_env(env, r, n, port), stdout=subprocess.PIPE if r == 0 else err,
                                      stderr=err, text=True))
    t0 = time.time()
    first_bad, codes = None, [None] * n
    relay = []
    try:
        import threading

        def pump():
            for line in procs[0].stdout:
                relay.append(line)
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
                    if codes[i] not in (None, 0) and first_bad is None:
                        first_bad = time.time()
                        print(f"[launch_ranks] rank {i} exited with {codes[i]}", file=err, flush=True)
            late = timeout is not None and time.time() - t0 > timeout
            if late or (first_bad is not None and time.time() - first_bad > grace):
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.terminate()
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        try:
                            p.wait(10)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            p.wait()
                        codes[i] = E_TIMEOUT if late else 0      # (stopped from here because another rank died: that rank's code is the job's)
                break
            time.sleep(0.05)
        th.join(10)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    out.write("".join(relay))
    out.flush()
    worst = 0
    for c in codes:
        c = 0 if c is None else c
        if c != 0 and (worst == 0 or abs(c) > abs(worst)):
            worst = c
    return (worst if worst >= 0 else 128 - worst), bool(relay)          # (a rank killed by signal s: 128 + s, as a shell reports it)
