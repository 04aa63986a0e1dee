#!/usr/bin/env python3
"""Issue-class mix of the hot loops of the kernels that carry the frame, for bench.py's `roofline.compute`.

    python tools/isa_mix.py profiles/r03_isa_mix.json

Compiles rt_api.hip to gfx950 assembly with the flags of the Makefile, takes the largest loop of each kernel (for k_light: one
light sample) and sorts its VALU instructions into the three issue classes measured by tools/valu_rate.hip
(profiles/r02_valu_rate.txt, 4 waves per SIMD, all lanes, nanoseconds per wave64 instruction and SIMD x the nominal 2.4 GHz):
    fast   2.7   v_fma/mul/add/sub_f32, v_mov_b32, v_add/sub_u32, v_and/or/xor_b32, v_lshrrev_b32
    trans  8.25  v_rcp/sqrt/rsq_f32 (16.5 for f64)
    slow   4.4   everything else: compares, selects, min/max(3), conversions, ldexp, left shifts, bit-field and three-operand integer
                 ops, integer multiplies, packed and f64 arithmetic, DPP moves, lane reads
Those are the costs of long runs of ONE kind of instruction.  In a mixed stream a slow-class instruction next to fast-class ones costs no more than
a fast one (same file: PAIR fma | max3 3.14 per instruction against 3.08 and 4.51 alone, PAIR fma | cvt_ubyte 3.15, TRIPLE fma mul | max3 2.45; two
slow ones next to each other stay slow: PAIR max3 | cvt_ubyte 4.42), while a transcendental adds its full cost (PAIR fma | rcp 6.06 = (3.08 + 8.25) / 2
+ 7 %).  So the issue ceiling of a mix is modelled with two limits, the issue port at the fast rate for every non-transcendental instruction and the
slow class's own pipe, plus the transcendentals:
    cycles = max((fast + slow) x 2.7, slow x 4.4) + trans x 8.25 + trans64 x 16.5
and the ceiling is 1024 SIMDs x 2.4e9 x instructions / cycles wave-instructions per second.  (Round 2 and the first half of round 3 added the
classes' costs up, which gave 679 G/s for k_light; k_light then measured 774 G/s once its grid let the dispatcher balance it, so the sum was no
ceiling.  `serial_sum_cycles_per_valu_instruction` keeps that figure for comparison.)
The count is static (one pass through the loop, cold paths included), so the ceiling is a model, not a measurement; the achieved rate
that bench.py sets against it is measured (SQ_INSTS_VALU per launch from the PMC passes over the launch time).
The file is stamped with the hash of the device sources; bench.py ignores it when the hash differs from the tree it runs.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FAST = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_mov_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_not_b32")
TRANS = ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")
TRANS64 = ("v_rcp_f64", "v_sqrt_f64", "v_rsq_f64")
CYCLES = {"fast": 2.7, "slow": 4.4, "trans": 8.25, "trans64": 16.5}
KERNELS = {"k_light": "_ZN2rt7k_lightILb0ELi4EEE", "k_trace_p": "_ZN2rt9k_trace_pILb0ELi6EEE", "k_extend_p": "_ZN2rt10k_extend_pILb0EEE", "k_shade_b": "_ZN2rt9k_shade_bE"}


def issue_class(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in TRANS64:
        return "trans64"
    if base in TRANS:
        return "trans"
    if base in FAST:
        return "fast"
    return "slow"


def main(out):
    from isa_budget import kernel_body
    from rustracer_amd._lib import kernel_source_hash
    csrc = os.path.join(ROOT, "rustracer_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "rt_api.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                               "--cuda-device-only", "-S", os.path.join(csrc, "rt_api.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        res = {}
        for name, prefix in KERNELS.items():
            _, body = kernel_body(asm, prefix)
            insts, labels = [], {}
            for l in body:
                t = l.strip()
                m = re.match(r"^(\.LBB[0-9_]+):", l)
                if m:
                    labels[m.group(1)] = len(insts)
                    continue
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                insts.append((t.split()[0], t))
            loops = []
            for i, (op, t) in enumerate(insts):
                if op.startswith(("s_cbranch", "s_branch")) and t.split()[-1] in labels and labels[t.split()[-1]] <= i:
                    loops.append((labels[t.split()[-1]], i))
            lo, hi = max(loops, key=lambda lp: lp[1] - lp[0]) if loops else (0, len(insts) - 1)
            mix = {"fast": 0, "slow": 0, "trans": 0, "trans64": 0}
            for op, _ in insts[lo:hi + 1]:
                if op.startswith("v_"):
                    mix[issue_class(op)] += 1
            n = sum(mix.values())
            serial = sum(mix[c] * CYCLES[c] for c in mix) / max(n, 1)
            cyc = max((mix["fast"] + mix["slow"]) * CYCLES["fast"], mix["slow"] * CYCLES["slow"]) + mix["trans"] * CYCLES["trans"] + mix["trans64"] * CYCLES["trans64"]
            mean = cyc / max(n, 1)
            res[name] = {"symbol_prefix": prefix, "loop_valu_instructions": n, "mix": mix, "mean_cycles_per_valu_instruction": round(mean, 3),
                         "serial_sum_cycles_per_valu_instruction": round(serial, 3),
                         "issue_peak_wave_instructions_per_s": round(1024 * 2.4e9 / mean, 1)}
    j = {"what": "static issue-class mix of the largest loop of each kernel (tools/isa_mix.py); cycles per class from profiles/r02_valu_rate.txt at 4 waves per SIMD; "
                 "ceiling = max((fast + slow) x c_fast, slow x c_slow) + trans x c_trans cycles per pass of the loop (mixed-stream rows of the same file)",
         "cycles_per_class": CYCLES, "simds": 1024, "nominal_clock_hz": 2.4e9, "kernel_source_hash": kernel_source_hash(), "kernels": res}
    open(out, "w").write(json.dumps(j, indent=1) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
