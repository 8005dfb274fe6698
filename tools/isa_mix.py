#!/usr/bin/env python3
"""Instruction mix of the per-entry loops of the two blend kernels, from the compiler's own assembly (no GPU needed):

    python tools/isa_mix.py > profiles/r04/isa_mix_render_v2.txt

Compiles ad-gs_amd/csrc/render_v2.hip for gfx950 with the flags of the Makefile, takes render_fwd_v2_kernel<4> and
render_bwd_v2_kernel<4, true>, finds in each the innermost loop that evaluates alpha (the one with v_exp_f32) and prints, per basic
block of that loop, the instruction classes that matter for the issue model measured in tools/microbench/issue_hazards.hip:
full-rate fp32 (fma / mul / add / sub / mov), packed fp32, half-rate (v_cmp, v_cndmask, v_min / v_max, DPP), quarter-rate (v_exp, v_rcp),
LDS, vector memory, scalar ALU, waits and branches.  The blocks are labelled by what they contain (header = the alpha evaluation,
strip = one 16x4 strip body, reduction = the LDS round trip of the backward's 14 sums)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ad-gs_amd", "csrc", "render_v2.hip")
KERNELS = [("render_fwd_v2_kernel<4>", "_ZN4adgs12_GLOBAL__N_120render_fwd_v2_kernelILi4EEEvNS_15RenderV2FwdArgsE"),
           ("render_bwd_v2_kernel<4, true>", "_ZN4adgs12_GLOBAL__N_120render_bwd_v2_kernelILi4ELb1EEEvNS_15RenderV2BwdArgsE")]

CLASSES = [
    ("quarter-rate (v_exp, v_rcp, v_log, v_sqrt)", re.compile(r"^v_(exp|rcp|log|sqrt|rsq)_")),
    ("half-rate compare (v_cmp*)", re.compile(r"^v_cmpx?_")),
    ("half-rate select (v_cndmask)", re.compile(r"^v_cndmask")),
    ("half-rate min/max/med3", re.compile(r"^v_(min|max|med3)")),
    ("half-rate DPP / lane ops", re.compile(r".*_dpp$|^v_readlane|^v_readfirstlane|^v_writelane|^v_permlane")),
    ("packed fp32 (v_pk_*)", re.compile(r"^v_pk_")),
    ("full-rate fp32 (fma, fmac, mul, add, sub)", re.compile(r"^v_(fma|fmac|mul|add|sub|subrev)_f32")),
    ("other VALU (mov, integer, shifts)", re.compile(r"^v_")),
    ("LDS read", re.compile(r"^ds_read")),
    ("LDS write", re.compile(r"^ds_write")),
    ("vector memory (global_*, buffer_*)", re.compile(r"^(global|buffer|flat)_")),
    ("s_waitcnt", re.compile(r"^s_waitcnt")),
    ("branches (s_cbranch, s_branch)", re.compile(r"^s_c?branch")),
    ("other SALU", re.compile(r"^s_")),
]


def classify(mn):
    for name, rx in CLASSES:
        if rx.match(mn):
            return name
    return "other"


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-fno-slp-vectorize",
               "-S", "--cuda-device-only", "-o", out, SRC] + sys.argv[1:]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    print("# instruction mix of the per-entry loops, hipcc flags: %s" % " ".join(cmd[1:7] + sys.argv[1:]))
    for title, sym in KERNELS:
        start = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
        end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
        body = text[start:end]
        meta = [l.strip() for l in text[end:end + 400] if re.search(r"NumVgprs:|NumSgprs:|Occupancy:|LDSByteSize:", l)][:4]
        # basic blocks
        blocks, cur, name = [], [], "entry"
        depth2 = {}
        for l in body:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                blocks.append((name, cur)); name, cur = m.group(1), []
                depth2[name] = "Depth=2" in l or "Depth=3" in l
                continue
            if "Loop Header: Depth=" in l and cur == []:
                depth2[name] = depth2.get(name, False) or "Depth=2" in l or "Depth=3" in l
            t = l.split(";")[0].strip()
            if not t or t.startswith("."):
                continue
            cur.append(t.split()[0])
        blocks.append((name, cur))
        # the loop of interest: the contiguous run of blocks around the first v_exp_f32 whose comments say Depth=2 (fall-through blocks "%bb." have no label)
        idx_exp = [i for i, (n, ins) in enumerate(blocks) if any(x.startswith("v_exp_f32") for x in ins)]
        print("\n## %s   (%s)" % (title, "; ".join(meta)))
        if not idx_exp:
            print("no v_exp_f32 found"); continue
        # walk outwards while blocks belong to a depth >= 2 loop
        lo = hi = [i for i in idx_exp if depth2.get(blocks[i][0], False)][0] if any(depth2.get(blocks[i][0], False) for i in idx_exp) else idx_exp[0]
        while lo - 1 >= 0 and depth2.get(blocks[lo - 1][0], False):
            lo -= 1
        while hi + 1 < len(blocks) and depth2.get(blocks[hi + 1][0], False):
            hi += 1
        total = collections.Counter()
        for n, ins in blocks[lo:hi + 1]:
            if not ins:
                continue
            c = collections.Counter(classify(x) for x in ins)
            role = []
            if c["quarter-rate (v_exp, v_rcp, v_log, v_sqrt)"] and any(x.startswith("v_exp") for x in ins):
                role.append("header: %d alpha evaluations" % sum(x.startswith("v_exp") for x in ins))
            if any(x.startswith("v_rcp") for x in ins):
                role.append("backward strip body")
            if c["LDS write"] >= 4:
                role.append("reduction of the 14 sums")
            if any(x.startswith("global_atomic") for x in ins):
                role.append("atomic")
            if not role and c["half-rate select (v_cndmask)"] >= 3:
                role.append("forward strip body")
            valu = sum(v for k, v in c.items() if "rate" in k or "VALU" in k or "packed" in k)
            print("%-12s %3d instructions, %3d VALU  %-44s %s" % (n, len(ins), valu, "[" + ", ".join(role) + "]" if role else "",
                                                                   ", ".join("%s %d" % (k.split(" (")[0], v) for k, v in c.most_common())))
            total.update(c)
        print("loop total (every path once): " + ", ".join("%s %d" % (k.split(" (")[0], v) for k, v in total.most_common()))
    print("\n# issue model (tools/microbench/issue_hazards.hip, profiles/r04/issue_hazards_baseline.txt; cycles per wave64 instruction and SIMD at 8 waves, wall):")
    print("# full-rate fp32 2.3, packed fp32 4.2 (two FMAs), half-rate classes 4.2-4.4, quarter-rate 8.2, SALU ~1.3 next to VALU, one wave alone: ~5.4 per instruction of any class")


if __name__ == "__main__":
    main()
