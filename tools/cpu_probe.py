import os, sys, time, ctypes as C, json, subprocess
sys.path.insert(0, 'mp3-steganography-lib_amd'); sys.path.insert(0, 'tests')
from mp3stego import _lib
from synth_pcm import synth_pcm
L = _lib.lib()
ctx = _lib.Context(0)
mp3 = bytes(ctx.encode_pcm(synth_pcm(10000, seed=1), 44100, 128, None)["mp3"])
buf = C.create_string_buffer(mp3, len(mp3))
print(subprocess.run("lscpu | egrep 'Model name|Socket|NUMA|Thread|MHz'; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null", shell=True, capture_output=True, text=True).stdout)
def t_scan():
    best = 1e9
    for i in range(5):
        o = C.c_void_p(); sc = _lib.Scanned()
        t = time.perf_counter(); L.mp3s_scan_stream(buf, len(mp3), C.byref(o), C.byref(sc)); dt = time.perf_counter() - t
        L.mp3s_buf_free(o); best = min(best, dt)
    return round(best * 1e3, 3)
print("free", t_scan())
for cpu in (0, 1, 2, 16, 32, 48, 64, 96, 127, 128, 129, 160, 192, 224, 255):
    try:
        os.sched_setaffinity(0, {cpu})
        print(cpu, t_scan())
    except Exception as e:
        print(cpu, "err", e)
