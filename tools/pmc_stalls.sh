#!/bin/bash
# Per-round stall breakdown of the round kernel: three PMC passes over one
# sequential-launch step (the kernels of the two launches not mixed), summarised per dispatch.
#   gpurun -- 'tools/pmc_stalls.sh gpurun_out/r04_stalls'     [SOIL_LIB selects the build]
out=/root/repo/${1:-gpurun_out/r04_stalls}; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sequential-particles"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/a -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/b -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $out/c -o p -- $B > /dev/null 2>&1
cd /root/repo
python - <<PY > $out/summary.txt
import csv, glob, collections, json
res = {}
for sub in "abc":
    fs = glob.glob("$out/%s/**/p_counter_collection.csv" % sub, recursive=True)
    ts = glob.glob("$out/%s/**/p_kernel_trace.csv" % sub, recursive=True)
    if not fs or not ts:
        print("pass", sub, "left no table"); continue
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(ts[0]))}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if "k_tiled_round" not in r["Kernel_Name"]: continue
        kind = "fluvial" if "k_tiled_round<0" in r["Kernel_Name"] else "debris"
        e = per.setdefault((int(r["Dispatch_Id"]), kind), {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (did, kind), e in per.items():
        t = trace.get(str(did))
        if t: e["dur_us"] = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    for kind in ("fluvial", "debris"):
        ds = [e for (did, k), e in sorted(per.items()) if k == kind]
        ds = ds[len(ds) // 2:]   # the second (timed) step
        for i, e in enumerate(ds):
            res.setdefault(kind, {}).setdefault(i, {}).update({k: v for k, v in e.items()})
for kind, rounds in res.items():
    print(kind)
    for i, e in sorted(rounds.items()):
        w = e.get("SQ_WAVE_CYCLES", 0) or 1
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
        line = ["round %2d %7.0f us" % (i, e.get("dur_us", 0))]
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"):
            if k in e: line.append("%s/wave=%.3f" % (k.replace("SQ_", ""), e[k] / w))
        if "SQ_BUSY_CYCLES" in e: line.append("waves/SIMD(avg)=%.2f" % (w * 4 / (e["SQ_BUSY_CYCLES"] * 4) if e["SQ_BUSY_CYCLES"] else 0))
        if "SQ_WAVES" in e: line.append("waves=%d" % e["SQ_WAVES"])
        if "SQ_INSTS_VALU" in e:
            n = e["SQ_INSTS_VALU"]
            line.append("VALU=%.3gG lanes=%.3f salu/valu=%.2f br/valu=%.3f lds/valu=%.3f cyc/valu(SIMD)=%.2f ldsbusy=%.3f conflict=%.3f" % (
                n / 1e9, e.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * 4 * n) if n else 0,   # calibrate the x4 against round 0
                e.get("SQ_INSTS_SALU", 0) / n, e.get("SQ_INSTS_BRANCH", 0) / n, e.get("SQ_INSTS_LDS", 0) / n,
                1024 * cyc / n, e.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), e.get("SQ_LDS_BANK_CONFLICT", 0) / (256 * cyc)))
        print("  " + "  ".join(line))
json.dump(res, open("$out/per_round.json", "w"), indent=1)
PY
cat $out/summary.txt
du -sh $out
