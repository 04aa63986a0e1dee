#!/usr/bin/env python3
"""ISA-level instruction budget of one kernel: classes of instructions in its hottest loop and in the whole body.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S \
        rustracer_amd/csrc/rt_api.hip -o /tmp/rt_api.s
  python tools/isa_budget.py /tmp/rt_api.s _ZN2rt7k_lightILb0ELi4ELb0EEE [--loop]

The inner loop is taken to be the innermost backward branch region with the most instructions (for k_light: one light
sample).  Counts are static (per wave, one pass through the loop); divergent regions are executed by a wave whenever
any lane needs them.
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        if op.endswith("_f64") or "_f64_" in op:
            if "div" in op or "rcp" in op or "sqrt" in op or "rsq" in op:
                return "VALU f64 div/sqrt group"
            if op.startswith("v_cvt"):
                return "VALU cvt"
            return "VALU f64 arithmetic"
        if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp")):
            return "VALU f32 divide group (div_scale/rcp/div_fmas/div_fixup)"
        if op.startswith(("v_sqrt", "v_rsq")):
            return "VALU f32 sqrt/rsq"
        if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_i64_i32")):
            return "VALU integer multiply (LCG)"
        if op.startswith(("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_pk_fma_f32")):
            return "VALU f32 fma"
        if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mac_f32")):
            return "VALU f32 mul/add"
        if op.startswith("v_cmp") or op.startswith("v_cndmask") or op.startswith("v_cmpx"):
            return "VALU compare/select"
        if op.startswith("v_cvt") or op.startswith(("v_rndne", "v_floor", "v_trunc", "v_ldexp", "v_frexp")):
            return "VALU cvt/round"
        if op.startswith(("v_max", "v_min", "v_med3")):
            return "VALU min/max"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
            return "VALU move"
        return "VALU other (int/logic/shift)"
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load")):
            return "SMEM load"
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            return "wait/nop"
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch"
        return "SALU"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "VMEM load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "VMEM store"
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "VMEM atomic"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


def kernel_body(path, prefix):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith(prefix) and l.split(":")[0].startswith(prefix) and ":" in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel %s not found" % prefix)
    body = []
    for l in lines[start + 1:]:
        if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return lines[start].split(":")[0], body


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    name, body = kernel_body(path, prefix)
    insts = []      # (index, label or None, op, text)
    labels = {}
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB[0-9_]+):", l)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = t.split()[0]
        insts.append((op, t))
    # backward branches = loops
    loops = []
    for i, (op, t) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    print("kernel %s: %d instructions, %d loops" % (name, len(insts), len(loops)))
    total = Counter(classify(op) for op, _ in insts)
    # innermost = loops that contain no other loop; take the largest of those
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    hot = max(inner, key=lambda lp: lp[1] - lp[0]) if inner else None
    # the loop that encloses `hot` most tightly but contains everything per sample is usually `hot` itself; also print the largest loop
    for title, lp in (("largest innermost loop", hot), ("largest loop", max(loops, key=lambda lp: lp[1] - lp[0]) if loops else None)):
        if lp is None:
            continue
        c = Counter(classify(op) for op, _ in insts[lp[0]:lp[1] + 1])
        n = lp[1] - lp[0] + 1
        valu = sum(v for k, v in c.items() if k.startswith("VALU"))
        print("\n%s: instructions [%d, %d] = %d, VALU %d" % (title, lp[0], lp[1], n, valu))
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print("  %-62s %5d" % (k, v))
    valu = sum(v for k, v in total.items() if k.startswith("VALU"))
    print("\nwhole kernel: VALU %d" % valu)
    for k, v in sorted(total.items(), key=lambda kv: -kv[1]):
        print("  %-62s %5d" % (k, v))


if __name__ == "__main__":
    main()
