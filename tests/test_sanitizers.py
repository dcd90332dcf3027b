"""Host side of the library under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available): the parser,
scanner, reveal parse and WAV parse over mutated streams, and the file / message code over random inputs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_under_asan_ubsan(mlib, golden_dir, tmp_path):
    import test_fuzz as tf
    out = tmp_path / "bin"
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host", "build.sh"), str(out)], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("g++ without sanitizer runtimes")
    assert r.returncode == 0, r.stderr[-2000:]
    data = open(os.path.join(golden_dir, "test.mp3"), "rb").read()
    corpus = np.load(os.path.join(golden_dir, "g7_decode_corpus.npz"))
    names = sorted({k.split("__")[0] for k in corpus.files})
    indir = tmp_path / "in"
    indir.mkdir()
    i = 0
    for seed in range(12):
        src = data if seed % 3 else corpus[names[seed % len(names)] + "__mp3"].tobytes()
        gen = tf.header_mutants(mlib, src, 20, seed) if seed % 2 else tf.mutants(src, 20, seed)
        for m in gen:
            if i % 5 == 0:                                           # cut somewhere inside the last frames
                m = m[:len(m) - int(np.random.default_rng(i).integers(1, 600))]
            (indir / f"{i:04d}.bin").write_bytes(m)
            i += 1
    for k, t in enumerate([b"", b"\xff", b"\xff\xfb", b"\xff\xfb\x90", b"\xff\xfb\x90\x00", b"ID3",
                           b"ID3\x03\x00\x00\x7f\x7f\x7f\x7f", b"RIFF", b"RIFFxxxxWAVEfmt \x10\x00\x00\x00\x01\x00\x02\x00"]):
        (indir / f"tiny{k}.bin").write_bytes(t)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(out / "parse_mutants"), str(indir)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "files" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert int(r.stdout.split("walked")[1].split()[0]) > 100, r.stdout           # the frame walk took (and matched the scan on) many of them
    r = subprocess.run([str(out / "files_messages"), "10"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "reveal bytes" in r.stdout, (r.stdout + r.stderr)[-3000:]
