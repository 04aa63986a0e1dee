#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory (rocprofv3 CSVs) into one markdown file for profiles/,
and optionally the per-launch HBM traffic of each kernel into a JSON that bench.py reads for `roofline.traffic`.

    python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>.md [profiles/r01_pmc.json width height light_samples bounces mesh]
"""
import json
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"rt::(k_[a-z_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


N_SIMD, N_XCD = 1024, 8


def valu_busy(a):
    return a.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / max(N_SIMD * a.get("GRBM_GUI_ACTIVE", 0) / N_XCD, 1)


def per_bounce(per_disp, lanes=1):
    """k_extend_p / k_light / k_trace_p run once per path level (bounce) and pass; a round of `lanes` passes is enqueued level by level across the
    lanes (lane 0's level l, lane 1's, ..., then level l + 1), so dispatch i of a kernel is level (i div lanes) mod levels."""
    out = ["## Per bounce (level = (dispatch index div %d render lanes) modulo the launches per pass; sums over the passes of the run)" % lanes, "",
           "| kernel | level | waves | VALU instructions | active lanes | VALU issue busy |", "|---|---|---|---|---|---|"]
    for k in sorted(per_disp):
        base = k.split("<")[0]
        if base not in ("k_extend_p", "k_light", "k_trace_p"):
            continue
        d = per_disp[k]
        need = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")
        if any(c not in d for c in need):
            continue
        n = len(d["SQ_WAVES"])
        per_pass = 8
        if n % (per_pass * lanes):
            continue
        rows = defaultdict(lambda: defaultdict(float))
        for c in need:
            for i, (_, v) in enumerate(sorted(d[c])):
                rows[(i // lanes) % per_pass][c] += v
        for lv in range(per_pass):
            r = rows[lv]
            out.append("| `%s` | %d | %d | %.4g | %.3f | %.3f |" % (k, lv, r["SQ_WAVES"], r["SQ_INSTS_VALU"],
                       r["SQ_THREAD_CYCLES_VALU"] / max(64.0 * r["SQ_ACTIVE_INST_VALU"], 1), valu_busy(r)))
    return out + [""]


def main(src, dst):
    out = ["# rocprofv3 summary: %s" % os.path.basename(src.rstrip("/")), ""]
    for line in (open(os.path.join(src, "trace.log")) if os.path.exists(os.path.join(src, "trace.log")) else []):
        if line.startswith("{"):
            out += ["bench line of the traced run:", "", "```", line.strip(), "```", ""]
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        out += ["## `rocprofv3 --kernel-trace --stats` (kernel_stats.csv)", "", "| kernel | calls | total ms | avg ms | % | min ms | max ms |",
                "|---|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(stats[0])):
            out.append("| `%s` | %s | %.3f | %.4f | %s | %.4f | %.4f |" % (short(r["Name"]) or r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                       float(r["AverageNs"]) / 1e6, r["Percentage"], float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
        out.append("")
    # PMC: sum each counter over the dispatches of each of our kernels
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    per_disp = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [(dispatch id, value)]
    meta = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
            per_disp[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    if agg:
        out += ["## PMC counters (separate `--pmc` passes; sums over all dispatches of the kernel in the run)", ""]
        counters = sorted({c for k in agg for c in agg[k]})
        out += ["| kernel | VGPR | SGPR | LDS B | scratch | " + " | ".join(counters) + " |", "|---|---|---|---|---|" + "---|" * len(counters)]
        for k in sorted(agg):
            m = meta[k]
            out.append("| `%s` | %s | %s | %s | %s | " % (k, m[0], m[2], m[3], m[4]) + " | ".join("%.4g" % agg[k].get(c, float("nan")) for c in counters) + " |")
        out += ["", "## Derived (per kernel)", "",
                "| kernel | dispatches | active lanes per VALU instruction | VALU issue busy | waves resident per SIMD | L2 hit rate | FETCH GB (x2 corrected) | WRITE GB |",
                "|---|---|---|---|---|---|---|---|"]
        for k in sorted(agg):
            a = agg[k]
            lane = a.get("SQ_THREAD_CYCLES_VALU", 0) / max(a.get("SQ_ACTIVE_INST_VALU", 0) * 64.0, 1)
            busy = valu_busy(a)
            occ = a.get("SQ_WAVE_CYCLES", 0) * 4.0 / max(N_SIMD * a.get("GRBM_GUI_ACTIVE", 0) / N_XCD, 1)
            hit = a.get("TCC_HIT_sum", 0) / max(a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0), 1)
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 FETCH_SIZE reads 1/2 of wide coalesced reads (MI355X_MICROARCH.md §HBM)
            fetch = a.get("FETCH_SIZE", 0) * 1024 * 2 / 1e9
            write = a.get("WRITE_SIZE", 0) * 1024 / 1e9
            out.append("| `%s` | %d | %.3f | %.3f | %.2f | %.4f | %.3f | %.3f |" % (k, calls[k].get("SQ_WAVES", 0), lane, busy, occ, hit, fetch, write))
        out += ["", "Notes: SQ_*_CYCLES / SQ_ACTIVE_* counters are in quad-cycles (MI355X_MICROARCH.md) and are summed over the chip; GRBM_GUI_ACTIVE is summed "
                "over the 8 XCDs.  'active lanes' = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): the mean fraction of the 64 lanes enabled per VALU "
                "instruction.  'VALU issue busy' = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): the fraction of all SIMD cycles of the kernel's "
                "run time in which the SIMD's VALU was executing (a wave64 instruction occupies its SIMD's 16-lane VALU for 4 cycles; counters of different "
                "passes come from different runs, so a ratio can exceed 1 by a few percent).  'waves resident' = 4 * SQ_WAVE_CYCLES over the same denominator.", ""]
        lanes = 1
        for line in (open(os.path.join(src, "trace.log")) if os.path.exists(os.path.join(src, "trace.log")) else []):
            if line.startswith("{"):
                lanes = int((json.loads(line).get("config") or {}).get("render_lanes") or 1)
        out += per_bounce(per_disp, lanes)
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))
    return agg, calls


def write_pmc_json(agg, calls, src, path, workload):
    """HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB / dispatches: FETCH_SIZE and WRITE_SIZE come from separate --pmc
    passes; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B-per-lane loads at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM)."""
    kernels = {}
    for k in agg:
        a = agg[k]
        if "FETCH_SIZE" not in a or "WRITE_SIZE" not in a:
            continue
        nf, nw = calls[k]["FETCH_SIZE"], calls[k]["WRITE_SIZE"]
        fetch = a["FETCH_SIZE"] * 1024 * 2 / max(nf, 1)
        write = a["WRITE_SIZE"] * 1024 / max(nw, 1)
        base = k.split("<")[0]
        e = kernels.setdefault(base, {"variants": [], "dispatches": 0, "hbm_bytes_per_launch": 0.0, "fetch_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0})
        # several template variants of one kernel: weight by dispatch count
        tot = e["dispatches"] + nf
        e["fetch_bytes_per_launch"] = (e["fetch_bytes_per_launch"] * e["dispatches"] + fetch * nf) / tot
        e["write_bytes_per_launch"] = (e["write_bytes_per_launch"] * e["dispatches"] + write * nf) / tot
        e["dispatches"] = tot
        e["variants"].append(k)
        if "TCC_HIT_sum" in a:
            e["l2_hit_rate"] = round(a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0), 1), 4)
        if "SQ_ACTIVE_INST_VALU" in a and "GRBM_GUI_ACTIVE" in a:
            e["valu_issue_busy"] = round(valu_busy(a), 4)
            e["valu_active_lanes"] = round(a.get("SQ_THREAD_CYCLES_VALU", 0) / max(64.0 * a["SQ_ACTIVE_INST_VALU"], 1), 4)
            e["valu_insts_per_launch"] = round(a.get("SQ_INSTS_VALU", 0) / max(calls[k].get("SQ_INSTS_VALU", 1), 1))
    # units per launch of the profiled workload (units.log: the same bench command with its instrumented pass), so that the figures travel per
    # UNIT (light sample, traced ray, intersect ray) and bench.py can scale them to the launch sizes of the run it reports
    units, run_cfg = {}, {}
    upath = os.path.join(src, "units.log")
    if os.path.exists(upath):
        for line in open(upath):
            if line.startswith("{"):
                b = json.loads(line)
                for kn, kv in ((b.get("roofline") or {}).get("kernels") or {}).items():
                    units[kn] = (kv["units_per_launch"], kv["unit"])
                run_cfg = {"spp": (b.get("config") or {}).get("spp"), "render_lanes": (b.get("config") or {}).get("render_lanes"), "max_paths": (b.get("config") or {}).get("max_paths")}
    for name, e in kernels.items():
        e["hbm_bytes_per_launch"] = round(e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"])
        e["fetch_bytes_per_launch"] = round(e["fetch_bytes_per_launch"])
        e["write_bytes_per_launch"] = round(e["write_bytes_per_launch"])
        if name in units and units[name][0] > 0:
            u = units[name][0]
            e["unit"] = units[name][1]
            e["units_per_launch_profiled"] = u
            e["hbm_bytes_per_unit"] = round(e["hbm_bytes_per_launch"] / u, 4)
            if "valu_insts_per_launch" in e:
                e["valu_insts_per_unit"] = round(e["valu_insts_per_launch"] / u, 5)      # wave64 instructions (SQ_INSTS_VALU) per unit
    j = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of tools/profile.sh, %s" % os.path.basename(src.rstrip("/")),
         "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-B requests as 64 B) + WRITE_SIZE KiB x 1024, averaged over the kernel's dispatches",
         "valu": "valu_issue_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); valu_active_lanes = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)",
         "workload": workload, "profiled_run": run_cfg, "kernels": kernels,
         "kernel_source_hash": (open(os.path.join(src, "source_hash.txt")).read().strip() if os.path.exists(os.path.join(src, "source_hash.txt")) else None),
         "git_commit": os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % os.path.dirname(os.path.abspath(__file__))).read().strip() or None}
    open(path, "w").write(json.dumps(j, indent=1) + "\n")


if __name__ == "__main__":
    agg_, calls_ = main(sys.argv[1], sys.argv[2])
    if len(sys.argv) >= 9:
        write_pmc_json(agg_, calls_, sys.argv[1], sys.argv[3],
                       {"width": int(sys.argv[4]), "height": int(sys.argv[5]), "light_samples": int(sys.argv[6]), "bounces": int(sys.argv[7]), "mesh": sys.argv[8]})
