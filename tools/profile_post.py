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
name = sys.argv[1] if len(sys.argv) > 1 else "r05_final"
src = os.path.join(ROOT, "gpurun_out", name)
dst = os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)


def targs(kernel):   # template arguments of k_tiled_round<KIND, DEP, TR, TC, NT, ALB, SPARSE, FAST>
    m = re.search(r"k_tiled_round<([^>]*)>", kernel)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def is_sparse(kernel):
    a = targs(kernel)
    return len(a) > 6 and a[6] in ("true", "1")


def is_fast(kernel):
    a = targs(kernel)
    return len(a) > 7 and a[7] in ("true", "1")


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
if os.path.isdir(os.path.join(src, "stats_default")):      # the default step: launches overlapped, lazy flux planes
    for f in os.listdir(os.path.join(src, "stats_default")):
        if f.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(src, "stats_default", f), os.path.join(dst, "kernel_stats_default_step.csv"))
for f in ("bench_line.json", "bench_line_exact.json", "bench_line_sequential.json", "bench_c2_1024x10000.json"):
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
# the default step's flavour of the fused cell kernel (flux planes not re-zeroed: 84 B/cell) and its
# storing first rounds come from the passes over the default bench
lazy_per = collections.defaultdict(dict)
if os.path.isdir(os.path.join(src, "fetch_lazy")):
    for d, key in (("fetch_lazy", "FETCH_SIZE"), ("write_lazy", "WRITE_SIZE")):
        acc = collections.defaultdict(list)
        for c in load(d):
            if key in c:
                acc[c["name"]].append(c[key])
        for k, v in acc.items():
            lazy_per[k][key + "_KB_avg"] = sum(v) / len(v)
            lazy_per[k]["launches_" + key.split("_")[0].lower()] = len(v)
    json.dump(lazy_per, open(os.path.join(dst, "pmc_fetch_write_per_kernel_default_step.json"), "w"), indent=1)
lazy_cells = next((v for k, v in lazy_per.items() if "k_erode_cells_fused" in k and k.rstrip(">").endswith("false")
                   and "FETCH_SIZE_KB_avg" in v and "WRITE_SIZE_KB_avg" in v), None)
cells = next((v for k, v in per.items() if "k_erode_cells_fused" in k), None)
if cells:
    bench = json.load(open(os.path.join(dst, "bench_line.json")))
    if lazy_cells and bench["roofline"].get("algorithmic_bytes_per_cell") == 84:
        cells = dict(lazy_cells, lazy=True)
    fetch = cells["FETCH_SIZE_KB_avg"] * 1024 * 2      # gfx950: FETCH_SIZE counts half the bytes
    write = cells["WRITE_SIZE_KB_avg"] * 1024
    json.dump({
        "kernel": "k_erode_cells_fused", "grid": bench["config"]["grid"], "lazy_flux": bool(cells.get("lazy")),
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
            "active_inst_valu_x4_per_cycle = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles) is NOT a busy "
            "share on gfx950: the counter ticks one quad-cycle per vector instruction whatever its issue "
            "cost and two per transcendental (valu_calibration.json) — kept for reference only; the "
            "roofline fraction of this kernel is priced from its instruction mix, "
            "profiles/particle_roofline.json.  A wave64 v_fma/add/mul occupies a SIMD for 2 cycles, "
            "compares, min/max, conversions, v_cndmask, DPP and the v_div_* helpers for 4, v_rcp/v_sqrt/"
            "v_exp for 8 (tools/microbench/valu_issue*.hip); cycles_per_valu_instruction = 1024 x cycles "
            "/ SQ_INSTS_VALU is wall time per instruction.  lane utilisation = "
            "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs "
            "x cycles), lds_bank_conflict likewise",
}
for kind, label in ((0, "fluvial_rounds"), (1, "debris_rounds")):
    # (round kernels queued ahead of the device's decision to stop return at once: not launches of a round)
    skipped = {c["id"] for c in valu if "k_tiled_round<%d" % kind in c["name"] and c.get("SQ_INSTS_VALU", 0) < 1e4}
    # (... and the sparse tiles' kernel of a round whose scan made none: a few us of work-groups that return)
    real = lambda c: "k_tiled_round<%d" % kind in c["name"] and c.get("dur_us", 0.0) >= 20.0
    ra = [c for c in valu if real(c)]
    rb = [c for c in lds if real(c)]
    ra, rb = ra[len(ra) // 2:], rb[len(rb) // 2:]
    rows = []
    for i, (a, b) in enumerate(zip(ra, rb)):
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        rows.append({
            "round": i, "kernel": "sparse tiles (one wave, hashed accumulators)" if is_sparse(a["name"]) else "dense tiles",
            "arithmetic": "fast" if is_fast(a["name"]) else "exact",
            "duration_us": round(a["dur_us"], 1), "shader_clock_ghz": round(cyc / a["dur_us"] / 1e3, 2),
            "SQ_INSTS_VALU": a["SQ_INSTS_VALU"], "SQ_INSTS_VALU_TRANS_F32": a["SQ_INSTS_VALU_TRANS_F32"],
            "active_inst_valu_x4_per_cycle": round(4 * a["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3),
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
# 4. the roofline of the kernel that carries the step (bench.py copies it into `roofline_particles`).
#    SQ_ACTIVE_INST_VALU is no measure of a busy pipe on gfx950: it ticks one quad-cycle per vector
#    instruction whatever its issue cost (2, 4 or ~16 cycles) and two per transcendental
#    (valu_calibration.json).  The issue time is therefore priced from the instruction mix — the
#    hardware's own type counters (pass `mix`) times the issue cost measured per opcode class
#    (tools/microbench/valu_issue*.hip).
COST = {"SQ_INSTS_VALU_ADD_F32": 2.0, "SQ_INSTS_VALU_MUL_F32": 2.0, "SQ_INSTS_VALU_FMA_F32": 2.0,
        "SQ_INSTS_VALU_TRANS_F32": 8.0, "SQ_INSTS_VALU_CVT": 4.0,
        "SQ_INSTS_VALU_INT32": 3.0,   # v_add/sub_u32 and shifts right 2, the rest 4
        "other": 3.5}                 # moves and logic 2; compares, selects, min/max, DPP, readlane 4
# ... and the two classes the counters do not resolve are priced from the opcodes that are in the
# stepping loop (tools/isa_mix.py compiles the kernel and reads the loop off the ISA)
isa = None
try:
    import subprocess
    isa = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py")]))
    json.dump(isa, open(os.path.join(dst, "isa_round_loop.json"), "w"), indent=1)
except Exception as e:          # no hipcc here: keep round 2's assumption
    print("isa_mix failed, 'other' stays at 3.5 cycles:", e)
roof = {}
mix_ok = os.path.isdir(os.path.join(src, "mix"))
mix = load("mix") if mix_ok else []
for kind, label in ((0, "fluvial_rounds"), (1, "debris_rounds")):
    rows = [c for c in mix if "k_tiled_round<%d" % kind in c["name"] and c.get("dur_us", 0.0) >= 20.0]
    rows = rows[len(rows) // 2:]
    if not rows:
        continue
    n = sum(r["SQ_INSTS_VALU"] for r in rows)
    typed = {k: sum(r.get(k, 0.0) for r in rows) for k in COST if k != "other"}
    other = n - sum(typed.values())
    cost = dict(COST)
    if isa and label in isa:
        cost["other"] = isa[label]["other_issue_cycles"]
        cost["SQ_INSTS_VALU_INT32"] = isa[label]["int32_issue_cycles"]
    issue = sum(typed[k] * cost[k] for k in typed) + other * cost["other"]
    cycles = sum(1024 * r["GRBM_GUI_ACTIVE"] / 8 for r in rows)
    roof[label] = {"valu_issue_share": round(issue / cycles, 3), "launches": len(rows),
                   "time_ms": round(sum(r["dur_us"] for r in rows) / 1e3, 3), "valu_instructions": n,
                   "issue_cycles_per_instruction": round(issue / n, 2),
                   "issue_cost_cycles": cost,
                   "wall_simd_cycles_per_instruction": round(cycles / n, 2),
                   "mix": {k.replace("SQ_INSTS_VALU_", ""): round(v / n, 3) for k, v in typed.items()} | {
                       "other": round(other / n, 3)}}
if roof:
    tot_t = sum(v["time_ms"] for v in roof.values())
    json.dump({
        "kernel": "k_tiled_round (all launches of one 8192^2 step)", "bound": "valu-issue",
        "achieved": sum(v["valu_issue_share"] * v["time_ms"] for v in roof.values()) / tot_t,
        "peak": 1.0, "unit": "share of SIMD cycles taken by the issue of vector instructions",
        "per_kind": roof,
        "issue_cost_cycles": "per kind above: ADD/MUL/FMA 2, TRANS 8, CVT 4 (tools/microbench/valu_issue*.hip); INT32 and "
                             "the unclassified rest priced by the opcodes in the stepping loop's ISA "
                             "(profiles/%s/isa_round_loop.json, tools/isa_mix.py)" % name,
        "source": "profiles/%s: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 "
                  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE on `python bench.py --steps 1 --warmup 1` "
                  "(gpurun_out/%s/mix), priced with the wave64 issue cycles per opcode class measured by "
                  "tools/microbench/valu_issue.hip / valu_issue2.hip (valu_issue.txt); SQ_ACTIVE_INST_VALU is "
                  "not usable as a busy counter on gfx950 (valu_calibration.json)" % (name, name),
    }, open(os.path.join(ROOT, "profiles", "particle_roofline.json"), "w"), indent=1)
# 5. the graph / stencil / conditioning kernels: kernel stats of tools/bench_stencils.py and
#    tools/bench_accumulate.py, HBM traffic per launch next to the algorithmic bytes
for sub in ("stencils", "accumulate"):
    d = os.path.join(src, sub)
    if os.path.isdir(d):
        for f in os.listdir(d):
            if f.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(d, f), os.path.join(dst, "%s_kernel_stats.csv" % sub))
        txt = os.path.join(src, "bench_%s.txt" % sub)
        if os.path.exists(txt):
            shutil.copy(txt, os.path.join(dst, "bench_%s.txt" % sub))
if os.path.isdir(os.path.join(src, "stencils_fetch")):
    per = collections.defaultdict(dict)
    for d, key in (("stencils_fetch", "FETCH_SIZE"), ("stencils_write", "WRITE_SIZE")):
        acc = collections.defaultdict(list)
        for c in load(d):
            if key in c and "dur_us" in c:
                acc[c["name"]].append((c[key], c["dur_us"]))
        for k, v in acc.items():
            per[k][key + "_KB_avg"] = sum(x for x, _ in v) / len(v)
            per[k]["dur_us_avg_" + key.split("_")[0].lower()] = sum(t for _, t in v) / len(v)
    for k, v in per.items():
        if "FETCH_SIZE_KB_avg" in v and "WRITE_SIZE_KB_avg" in v:
            v["hbm_bytes_per_launch"] = v["FETCH_SIZE_KB_avg"] * 1024 * 2 + v["WRITE_SIZE_KB_avg"] * 1024
            v["achieved_GBs"] = v["hbm_bytes_per_launch"] / (v["dur_us_avg_fetch"] * 1e-6) / 1e9
    json.dump({"note": "tools/bench_stencils.py at 8192^2 under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two "
                       "passes); FETCH_SIZE x2 as in traffic_fused_cells.json; durations from the kernel trace "
                       "of the FETCH pass", "kernels": per},
              open(os.path.join(dst, "pmc_stencils_fetch_write.json"), "w"), indent=1)

# 6. calibration of valu_busy: the microbenchmark's opcode streams under the same counters
if os.path.isdir(os.path.join(src, "calib")):
    rows = []
    for c in load("calib"):
        if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            rows.append({"kernel": c["name"], "dur_us": round(c.get("dur_us", 0), 1),
                         "SQ_INSTS_VALU": c["SQ_INSTS_VALU"],
                         "active_inst_valu_x4_per_cycle": round(4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3),
                         "active_quads_per_instruction": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 3)})
    json.dump({"note": "tools/microbench/valu_issue: streams of one opcode each (k<MODE>; the order of the "
                       "launches is that of the program's output, valu_issue.txt; MODE = the enum of "
                       "valu_issue.hip: 0 fma 1 mul 2 add 3 pk_fma 4 pk_mul 5 pk_add 6 rcp 7 sqrt 8 exp 9 log "
                       "10 rsq 11 ldexp 12 floor 13 cvt 14 mad_u24 15 cndmask(vcc) 16 cmp 17 div_scale 18 div_fmas "
                       "19 div_fixup 20 max 21 med3 22 mov 23 dpp add 24 readlane 25 and).  SQ_ACTIVE_INST_VALU "
                       "reads one quad-cycle per instruction for every class (two for the transcendentals) — and "
                       "so more than 1.0 'busy' on a saturating 2-cycle stream: it cannot serve as a busy counter",
               "launches": rows}, open(os.path.join(dst, "valu_calibration.json"), "w"), indent=1)
    if os.path.exists(os.path.join(src, "valu_issue.txt")):
        shutil.copy(os.path.join(src, "valu_issue.txt"), os.path.join(dst, "valu_issue.txt"))
print("wrote", sorted(os.listdir(dst)))
