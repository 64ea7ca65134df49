"""Turns what tools/profile_final.sh left under gpurun_out/<name>/ into the summaries
kept under profiles/<name>/ (and profiles/traffic_fused_cells.json)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
name = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
src = os.path.join(ROOT, "gpurun_out", name)
dst = os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)


def short(kernel):
    k = re.sub(r"\(.*", "", kernel).replace("void ", "")
    return k


def load(d):
    """per dispatch: counters summed over the XCDs + duration from the kernel trace"""
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(src, d, "p_kernel_trace.csv")))}
    cnt = collections.defaultdict(dict)
    for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        c = cnt[r["Dispatch_Id"]]
        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        c["name"] = short(r["Kernel_Name"])
    out = []
    for did, c in cnt.items():
        t = trace.get(did)
        if t:
            c["dur_us"] = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
        c["id"] = int(did)
        out.append(c)
    out.sort(key=lambda c: c["id"])
    return out


# 1. kernel stats + bench lines
for f in os.listdir(os.path.join(src, "stats")):
    if f.endswith("kernel_stats.csv"):
        shutil.copy(os.path.join(src, "stats", f), os.path.join(dst, "kernel_stats.csv"))
for f in ("bench_line.json", "bench_c2_1024x10000.json"):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)) > 2:
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))

# 2. HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE are reported in KiB)
per = collections.defaultdict(dict)
for d, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for c in load(d):
        if key in c:
            acc[c["name"]].append(c[key])
    for k, v in acc.items():
        per[k][key + "_KB_avg"] = sum(v) / len(v)
        per[k]["launches_" + key.split("_")[0].lower()] = len(v)
json.dump(per, open(os.path.join(dst, "pmc_fetch_write_per_kernel.json"), "w"), indent=1)
cells = next((v for k, v in per.items() if "k_erode_cells_fused" in k), None)
if cells:
    bench = json.load(open(os.path.join(dst, "bench_line.json")))
    fetch = cells["FETCH_SIZE_KB_avg"] * 1024 * 2      # gfx950: FETCH_SIZE counts half the bytes
    write = cells["WRITE_SIZE_KB_avg"] * 1024
    json.dump({
        "kernel": "k_erode_cells_fused", "grid": bench["config"]["grid"],
        "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
        "source": "profiles/%s/pmc_fetch_write_per_kernel.json (rocprofv3 --pmc FETCH_SIZE and --pmc "
                  "WRITE_SIZE in separate passes of `python bench.py --steps 2 --warmup 1`)" % name,
        "corrections": "gfx950 FETCH_SIZE counts half of the fetched bytes: x2 (calibrated on "
                       "k_layers_from_planes, whose 256 MiB read is reported as 128 MiB); WRITE_SIZE x1 "
                       "(calibrated on k_noise, 256 MiB written); both counters are reported in KiB",
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    }, open(os.path.join(ROOT, "profiles", "traffic_fused_cells.json"), "w"), indent=1)

# 3. the round kernels: VALU issue / lane utilisation and LDS activity per launch (last step)
valu, lds = load("valu"), load("lds")
summary = {
    "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python bench.py --steps 1 "
               "--warmup 1 --no-cpu-baseline (8192^2, second step); two passes, see tools/profile_final.sh",
    "note": "counters are summed over the 8 XCDs; GRBM_GUI_ACTIVE/8 = shader cycles of the launch.  "
            "valu_busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles): the share of all SIMD cycles "
            "in which a vector instruction was executing (the counter ticks in quad-cycles) — the "
            "roofline fraction of this kernel, whose bound is VALU issue.  There is no single 'peak "
            "instructions per cycle': a wave64 v_fma/add/mul occupies a SIMD for 2 cycles, compares, "
            "min/max, conversions, v_cndmask, DPP and the v_div_* helpers for 4, v_rcp/v_sqrt/v_exp for "
            "8 (tools/microbench/valu_issue*.hip); cycles_per_valu_instruction = 1024 x cycles / "
            "SQ_INSTS_VALU is what the mix averages to including idle time.  lane utilisation = "
            "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs "
            "x cycles), lds_bank_conflict likewise",
}
for kind, label in ((0, "fluvial_rounds"), (1, "debris_rounds")):
    ra = [c for c in valu if "k_tiled_round<%d" % kind in c["name"]]
    rb = [c for c in lds if "k_tiled_round<%d" % kind in c["name"]]
    ra, rb = ra[len(ra) // 2:], rb[len(rb) // 2:]
    rows = []
    for i, (a, b) in enumerate(zip(ra, rb)):
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        rows.append({
            "round": i, "duration_us": round(a["dur_us"], 1), "shader_clock_ghz": round(cyc / a["dur_us"] / 1e3, 2),
            "SQ_INSTS_VALU": a["SQ_INSTS_VALU"], "SQ_INSTS_VALU_TRANS_F32": a["SQ_INSTS_VALU_TRANS_F32"],
            "valu_busy": round(4 * a["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3),
            "cycles_per_valu_instruction": round(1024 * cyc / a["SQ_INSTS_VALU"], 2),
            "SQ_ACTIVE_INST_VALU": a["SQ_ACTIVE_INST_VALU"], "shader_cycles": cyc,
            "lane_utilisation": round(a["SQ_THREAD_CYCLES_VALU"] / (64 * a["SQ_ACTIVE_INST_VALU"]), 3),
            "waves_per_cu": round(4 * a["SQ_WAVE_CYCLES"] / (cyc * 256), 1),
            "SQ_INSTS_SALU": b["SQ_INSTS_SALU"], "SQ_INSTS_BRANCH": b["SQ_INSTS_BRANCH"],
            "SQ_INSTS_LDS": b["SQ_INSTS_LDS"],
            "lds_busy": round(b["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 3),
            "lds_bank_conflict": round(b["SQ_LDS_BANK_CONFLICT"] / (256 * cyc), 3),
        })
    summary[label] = rows
json.dump(summary, open(os.path.join(dst, "pmc_round_kernel_valu.json"), "w"), indent=1)
# 4. the roofline of the kernel that carries the step (bench.py copies it into `roofline_particles`)
roof = {}
for label in ("fluvial_rounds", "debris_rounds"):
    rows = summary[label]
    busy = sum(4 * r["SQ_ACTIVE_INST_VALU"] for r in rows)
    total = sum(1024 * r["shader_cycles"] for r in rows)
    roof[label] = {"valu_busy": round(busy / total, 3), "launches": len(rows),
                   "time_ms": round(sum(r["duration_us"] for r in rows) / 1e3, 3),
                   "valu_instructions": sum(r["SQ_INSTS_VALU"] for r in rows)}
json.dump({
    "kernel": "k_tiled_round (all launches of one 8192^2 step)", "bound": "valu-issue",
    "achieved": sum(v["valu_busy"] * v["time_ms"] for v in roof.values()) / sum(v["time_ms"] for v in roof.values()),
    "peak": 1.0, "unit": "share of SIMD cycles executing a vector instruction",
    "per_kind": roof,
    "source": "profiles/%s/pmc_round_kernel_valu.json (rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU "
              "SQ_ACTIVE_INST_VALU ... GRBM_GUI_ACTIVE on `python bench.py --steps 1 --warmup 1`)" % name,
    "issue_cost_model": "tools/microbench/valu_issue.hip, valu_issue2.hip: wave64 issue cycles per opcode "
                        "class measured on this chip (2 / 4 / 8)",
}, open(os.path.join(ROOT, "profiles", "particle_roofline.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
