"""What the probes of k_rate_loop end on, and the shader clocks of a wave's phases (library built with -DMP3S_RL_STATS=1:
tools/build_variant.sh rlstats -DMP3S_RL_STATS=1; the kernel adds into g_rl_stats, read through mp3s_debug_rl_stats).
usage (GPU box): python tools/rl_stats.py [variant.so] > gpurun_out/rl_stats.json"""
import ctypes, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mp3-steganography-lib_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mp3stego import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
else:
    _lib.LIB_PATH = os.path.join(ROOT, "mp3-steganography-lib_amd", "build", "ab", "rlstats.so")
from synth_pcm import synth_pcm
n = 10000
ctx = _lib.Context(0)
ctx.set_option("file_pipeline", 0)            # one full-size launch per stage, as in the resident step
mp3 = bytes(ctx.encode_pcm(synth_pcm(n, seed=0x9E3779B97F4A7C15), 44100, 128, None)["mp3"])
L = _lib.lib()
buf = np.zeros(128, dtype=np.uint64)
L.mp3s_debug_rl_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
ctx.hide_message(mp3, "abcdefghijklmnopqrstuvwxyz0123456789")
assert L.mp3s_debug_rl_stats(buf.ctypes.data, 1) == 0
ctx.hide_message(mp3, "abcdefghijklmnopqrstuvwxyz0123456789")
assert L.mp3s_debug_rl_stats(buf.ctypes.data, 1) == 0
d = buf.astype(np.int64)
units, waves = int(d[35]), int(d[60])
names = ["precheck", "quantize refused", "lower bound", "upper bound", "in full"]
out = {"waves": waves, "active_units": units, "probes": {}, "inner": {}, "phase_clocks_per_wave": {}}
for j in range(7):
    row = d[5 * j:5 * j + 5]
    out["probes"]["probe %d" % (j + 1)] = {names[k]: round(float(row[k]) / units, 4) for k in range(5)}
    out["probes"]["probe %d" % (j + 1)]["too many bits"] = round(float(d[40 + j]) / units, 4)
    out["probes"]["probe %d" % (j + 1)]["of those the thresholds alone decide"] = round(float(d[47 + j]) / max(int(d[40 + j]), 1), 4)
out["inner"] = {"evaluations per unit": round(float(d[36]) / units, 4), "last probe reused": round(float(d[37]) / units, 4),
                "ended on the lower bound": round(float(d[38]) / units, 4), "in full": round(float(d[39]) / units, 4)}
out["evaluations per unit (quantised)"] = round(float(d[62]) / units, 4)
out["evaluations in full whose three regions all stay below 15"] = round(float(d[64]) / max(int(d[63]), 1), 4)
out["evaluations in full whose regions 1 and 2 stay below 15"] = round(float(d[65]) / max(int(d[63]), 1), 4)
out["evaluations with quadruples"] = round(float(d[61]) / max(int(d[62]), 1), 4)
for k, nm in enumerate(["front (lines, energies)", "tables into LDS + barrier", "first probe (pre-check)", "binary search", "inner loop", "results out"]):
    out["phase_clocks_per_wave"][nm] = round(float(d[54 + k]) / waves, 1)
print(json.dumps(out, indent=1))
