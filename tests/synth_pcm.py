"""Deterministic synthetic PCM for the benchmark/test workloads.

xorshift64* noise + two tones + a sweep, 44.1 kHz stereo int16.  Pure numpy so the
same samples are produced in the build container and on the GPU box.
"""
import numpy as np

SEED = 0x9E3779B97F4A7C15


def xorshift64star(n, seed=SEED):
    """n uint64 outputs of xorshift64* (vectorised in blocks by jumping through the scalar recurrence)."""
    out = np.empty(n, dtype=np.uint64)
    x = int(seed) & 0xFFFFFFFFFFFFFFFF
    if x == 0:
        x = 1
    M = 0xFFFFFFFFFFFFFFFF
    mult = 0x2545F4914F6CDD1D
    for i in range(n):
        x ^= x >> 12
        x ^= (x << 25) & M
        x ^= x >> 27
        out[i] = (x * mult) & M
    return out


def _noise(n, seed):
    """uniform [-1, 1): one xorshift64* state per 4096-sample block (fast: seeds a numpy PCG from it)."""
    blocks = (n + 4095) // 4096
    seeds = xorshift64star(blocks, seed)
    out = np.empty(blocks * 4096)
    for b in range(blocks):
        out[b * 4096:(b + 1) * 4096] = np.random.Generator(np.random.PCG64(int(seeds[b]))).random(4096)
    return out[:n] * 2 - 1


def synth_pcm(n_frames, seed=SEED, rate=44100):
    """int16 [n_frames*1152][2]: L = 0.25 sin(2pi 440 t) + 0.15 sin(2pi (1000+3000 tri(t/5)) t) + noise,
    R = 0.7 L + 0.1 sin(2pi 660 t) + independent noise; noise uniform +-0.02."""
    n = n_frames * 1152
    t = np.arange(n, dtype=np.float64) / rate
    tri = 2 * np.abs((t / 5.0) % 1.0 - 0.5)
    left = 0.25 * np.sin(2 * np.pi * 440 * t) + 0.15 * np.sin(2 * np.pi * (1000 + 3000 * tri) * t)
    right = 0.7 * left + 0.1 * np.sin(2 * np.pi * 660 * t)
    left = left + 0.02 * _noise(n, seed)
    right = right + 0.02 * _noise(n, seed ^ 0xD1B54A32D192ED03)
    pcm = np.stack([left, right], axis=1)
    return np.clip(np.rint(pcm * 32767), -32768, 32767).astype(np.int16)
