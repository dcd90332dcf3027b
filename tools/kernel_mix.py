#!/usr/bin/env python3
"""Static instruction mix of the step kernels, by the issue-cost classes tools/ubench/issue_rates.hip measures (CPU: compiles the device code
to assembly).  Writes profiles/r06_kernel_mix.json; bench.py weights the dominant kernel's vector instructions with the measured cost of
their class (profiles/r05_issue_rates.json) instead of a flat 4 clocks.  The mix is the LISTING's (every instruction counted once), which
stands for the dynamic one: the kernels' hot loops are unrolled straight-line code that makes up most of their listing.
usage: python tools/kernel_mix.py"""
import collections, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_stats

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"}


def cls(op, line):
    if not op.startswith("v_"):
        return None
    base = op
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        base = base.replace(suf, "")
    if base in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return "lane"
    if base in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return "mad64"
    if "dpp" in op or "row_" in line or "quad_perm" in line or "wave_sh" in line:
        return "half"
    if base in FULL and not op.endswith("_e64"):
        return "full"
    return "half"


if __name__ == "__main__":
    ks = isa_stats.kernels(isa_stats.listing())
    out = {}
    for name, ins in ks.items():
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        if not short.startswith("mp3s::k_"):
            continue
        c = collections.Counter()
        for l in ins:
            k = cls(l.split()[0], l)
            if k:
                c[k] += 1
        out[short.replace("mp3s::", "")] = dict(c, valu=sum(c.values()), instructions=len(ins))
    path = os.path.join(isa_stats.ROOT, "profiles", "r06_kernel_mix.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k in ("k_rate_loop", "k_enc_analysis", "k_dec_stream<2, false, false>", "k_enc_mdct", "k_enc_pack", "k_dec_huffman<4, 64>"):
        print(k, out.get(k))
