#!/usr/bin/env python3
"""Static look at the device code: per kernel the instruction count of its listing and the commonest opcodes.
usage: python tools/isa_stats.py [kernel-name-substring ...]   (compiles csrc/mp3s_device.hip to assembly, a minute)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mp3-steganography-lib_amd")

def listing(path=None):
    out = path or os.path.join(tempfile.gettempdir(), "mp3s_dev.s")
    if not path or not os.path.exists(out):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                               "--cuda-device-only", "-S", "-o", out, os.path.join(PKG, "csrc", "mp3s_device.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()

def kernels(text):
    res = {}
    for m in re.finditer(r"^(_ZN4mp3s\w+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ins = [l.strip() for l in body.split("\n")]
        ins = [l for l in ins if l and not l.startswith((";", ".", "//")) and not l.split(";")[0].strip().endswith(":")]
        res[name] = ins
    return res

if __name__ == "__main__":
    ks = kernels(listing())
    pats = sys.argv[1:]
    for name, ins in ks.items():
        if pats and not any(p in name for p in pats):
            continue
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        c = collections.Counter(i.split()[0] for i in ins)
        cls = collections.Counter()
        for op, n in c.items():
            k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_barrier", "s_nop")) else \
                "smem" if op.startswith(("s_load", "s_buffer")) else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other"
            cls[k] += n
        print(f"{short}: {len(ins)} instructions  {dict(cls)}")
        if pats:
            print("   ", ", ".join(f"{op} {n}" for op, n in c.most_common(24)))
