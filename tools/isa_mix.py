#!/usr/bin/env python
"""Opcode mix of the stepping loop of k_tiled_round, read off the ISA hipcc emits.

The hardware's typed instruction counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _CVT) leave
a third of the round kernel's vector instructions unclassified ("other").  Round 2 priced that class at
an assumed 3.5 issue cycles; this prices it from the opcodes that are actually in the loop: the kernel
is compiled to assembly (the product's flags), the innermost loop that holds the step's exponentials
is taken as the stepping loop, and every vector instruction in it is put into the issue-cost classes
measured by tools/microbench/valu_issue*.hip (2 / 4 / 8 cycles per wave64 instruction and SIMD).

    python tools/isa_mix.py            -> JSON on stdout (per kind: histogram, class shares, cost of "other")
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASS_A = re.compile(r"^v_(fma_f32|fmac_f32|mad_f32|add_f32|sub_f32|subrev_f32|mul_f32|mov_b32|mov_b64|and_b32|or_b32|"
                     r"xor_b32|not_b32|lshrrev_b32|ashrrev_i32|add_u32|sub_u32|subrev_u32|add_co_u32|addc_co_u32)")
CLASS_C = re.compile(r"^v_(rcp|rsq|sqrt|exp|log)_f32")
TYPED = {
    "ADD_F32": re.compile(r"^v_(add|sub|subrev)_f32|^v_pk_add_f32"),
    "MUL_F32": re.compile(r"^v_mul_f32|^v_pk_mul_f32"),
    "FMA_F32": re.compile(r"^v_(fma|fmac|mad)_f32|^v_pk_fma_f32"),
    "TRANS_F32": CLASS_C,
    "CVT": re.compile(r"^v_cvt_"),
    "INT32": re.compile(r"^v_(add_u32|sub_u32|subrev_u32|add3_u32|lshl_add_u32|mul_lo_u32|mul_u32_u24|mad_u32_u24|"
                        r"mul_hi_u32|lshlrev_b32|lshrrev_b32|ashrrev_i32|add_co_u32|addc_co_u32|lshl_add_u64)"),
}


def cost(op):
    if CLASS_C.match(op):
        return 8
    if CLASS_A.match(op) and "dpp" not in op:
        return 2
    return 4


def stepping_loop(asm, kernel_re):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(kernel_re, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    labels, ins = {}, []
    for l in lines[start:end]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        ins.append(s.split(";")[0].strip())
    loops = []
    for k, s in enumerate(ins):
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            loops.append((labels[m.group(1)], k))
    # the stepping loop: the innermost loop that holds both the attenuations (v_exp_f32) and the
    # compare-and-swap deposits (the NaN walkers' loop beside it deposits with native adds)
    both = [(b - a, a, b) for a, b in loops
            if any(x.startswith("v_exp_f32") for x in ins[a:b + 1]) and any(x.startswith("ds_cmpst") for x in ins[a:b + 1])]
    _, a, b = min(both)
    return ins[a:b + 1]


def main():
    from soillib_amd import build
    src = os.path.join(ROOT, "soillib_amd", "csrc", "erosion_particles_tiled.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "tiled.s")
        subprocess.check_call([build._hipcc()] + build.FLAGS + ["--cuda-device-only", "-S", "-o", out, src],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    res = {}
    # the variant the bench line runs: fast arithmetic (SOIL_ISA_ARITH=exact: the IEEE step)
    fast = os.environ.get("SOIL_ISA_ARITH", "fast") != "exact"
    tail = "ELb0ELb0ELb%dEE" % (1 if fast else 0)     # ALB, SPARSE, FAST
    res_arith = "fast" if fast else "exact"
    for kind, label, pat in ((0, "fluvial_rounds", r"^_ZN4soil13k_tiled_roundILi0ELi1ELi78ELi64ELi768%s.*:" % tail),
                             (1, "debris_rounds", r"^_ZN4soil13k_tiled_roundILi1ELi1ELi104ELi64ELi768%s.*:" % tail)):
        loop = stepping_loop(asm, pat)
        valu = [i.split()[0] for i in loop if i.startswith("v_") and not i.startswith("v_readlane") or "dpp" in i]
        valu += [i.split()[0] for i in loop if i.startswith("v_readlane") or i.startswith("v_writelane")]
        hist = {}
        for op in valu:
            hist[op] = hist.get(op, 0) + 1
        typed = {k: sum(n for op, n in hist.items() if rx.match(op)) for k, rx in TYPED.items()}
        is_typed = lambda op: any(rx.match(op) for rx in TYPED.values())
        other_ops = {op: n for op, n in hist.items() if not is_typed(op)}
        n_other = sum(other_ops.values())
        res[label] = {
            "arithmetic": res_arith,
            "loop_instructions": len(loop), "valu": len(valu),
            "salu_and_waitcnt": sum(1 for i in loop if i.startswith("s_") and not i.startswith("s_cbranch") and not i.startswith("s_branch")),
            "branches": sum(1 for i in loop if i.startswith("s_cbranch") or i.startswith("s_branch")),
            "lds": sum(1 for i in loop if i.startswith("ds_")), "vmem": sum(1 for i in loop if i.startswith("global_")),
            "typed_static": typed, "other_static": n_other,
            "other_issue_cycles": round(sum(cost(op) * n for op, n in other_ops.items()) / max(n_other, 1), 3),
            "int32_issue_cycles": round(sum(cost(op) * n for op, n in hist.items() if TYPED["INT32"].match(op)) /
                                        max(typed["INT32"], 1), 3),
            "all_valu_issue_cycles": round(sum(cost(op) * n for op, n in hist.items()) / max(len(valu), 1), 3),
            "other_opcodes": dict(sorted(other_ops.items(), key=lambda kv: -kv[1])),
        }
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
