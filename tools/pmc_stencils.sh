#!/bin/bash
# What binds each per-cell kernel: two PMC passes over tools/bench_stencils.py,
# summarised per kernel — duration, vector instructions per cell, their issue time priced with the
# wave64 issue cycles per opcode class (tools/microbench/valu_issue.txt: add/mul/fma 2, transcendental 8,
# conversions 4, integer 3, everything else — compares, selects, DPP — 4) as a share of the SIMDs'
# cycles, and how long a wave waits.
#   gpurun -- 'tools/pmc_stencils.sh gpurun_out/r04_stencils'   [SOIL_WIN_DMA=0|1 picks the walk]
out=/root/repo/${1:-gpurun_out/r04_stencils}; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/tools/bench_stencils.py --reps 3"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $out/mix -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/wait -o p -- $B > /dev/null 2>&1
cd /root/repo
python - <<PY > $out/summary.txt
import csv, glob, collections, re
S = 8192
per = collections.OrderedDict()
for sub in ("mix", "wait"):
    fs = glob.glob("$out/%s/**/p_counter_collection.csv" % sub, recursive=True)
    ts = glob.glob("$out/%s/**/p_kernel_trace.csv" % sub, recursive=True)
    if not fs or not ts:
        print("pass", sub, "left no table"); continue
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(ts[0]))}
    disp = {}
    for r in csv.DictReader(open(fs[0])):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for did, e in disp.items():
        t = trace.get(did)
        if not t: continue
        name = re.sub(r"^void soil::|\(.*$", "", e["name"])
        a = per.setdefault(name, {"n": collections.Counter()})
        for k, v in e.items():
            if k == "name": continue
            a[k] = a.get(k, 0.0) + v; a["n"][k] += 1
        a["dur_" + sub] = a.get("dur_" + sub, 0.0) + (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
        a["n"]["dur_" + sub] += 1
COST = {"ADD_F32": 2, "MUL_F32": 2, "FMA_F32": 2, "TRANS_F32": 8, "INT32": 3, "CVT": 4}
print("%-58s %8s %7s %7s %7s %7s %7s" % ("kernel (8192^2, averages per launch)", "us", "valu/c", "issue", "wait", "salu/v", "lds/v"))
for name, a in per.items():
    g = lambda k: a.get(k, 0.0) / max(a["n"].get(k, 0), 1)
    n = g("SQ_INSTS_VALU")
    if n < 1e6 or "fill" in name or "noise" in name: continue
    cyc = g("GRBM_GUI_ACTIVE") / 8
    typed = sum(g("SQ_INSTS_VALU_" + k) for k in COST)
    issue = sum(g("SQ_INSTS_VALU_" + k) * c for k, c in COST.items()) + max(n - typed, 0) * 4
    w = g("SQ_WAVE_CYCLES") or 1
    print("%-58s %8.1f %7.1f %7.3f %7.3f %7.2f %7.3f" % (name[:58], g("dur_mix"), n * 64 / (S * S), issue / (1024 * cyc) if cyc else 0,
          g("SQ_WAIT_INST_ANY") / w, g("SQ_INSTS_SALU") / n, g("SQ_INSTS_LDS") / n))
print("valu/c: vector instructions per cell (wave instructions x 64 / cells; four cells per lane: x 1/4 of the lane's count)")
print("issue: priced issue cycles / (1024 SIMDs x shader cycles); wait: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES")
PY
cat $out/summary.txt
du -sh $out
