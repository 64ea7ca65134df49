#!/bin/bash
# Byte accounting of BASELINE config 3 (4096^2 D8 multiple-flow accumulation + pit fill): three passes
# over tools/bench_accumulate.py --k 8 (durations; FETCH_SIZE; WRITE_SIZE), summarised per kernel
# and, for k_rake_compress, per round of one accumulation.  FETCH_SIZE is corrected x2 as in
# tools/profile_post.py (calibrated on k_layers_from_planes); both counters are in KiB.
#   gpurun -- 'tools/pmc_accumulate.sh gpurun_out/r04_acc'
out=/root/repo/${1:-gpurun_out/r04_acc}; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/tools/bench_accumulate.py --k 8"
$B > $out/bench_accumulate_k8.txt 2>/dev/null
python /root/repo/tools/bench_accumulate.py --k 32 > $out/bench_accumulate_k32.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- $B > /dev/null 2>&1
cd /root/repo
python - <<PY > $out/bench_accumulate.txt
import csv, glob, collections, re
S = 4096; cells = S * S
def load(sub, key):
    f = glob.glob("$out/%s/**/p_counter_collection.csv" % sub, recursive=True)
    t = glob.glob("$out/%s/**/p_kernel_trace.csv" % sub, recursive=True)
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(t[0]))}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != key: continue
        e = per.setdefault(int(r["Dispatch_Id"]), {"name": re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), "v": 0.0})
        e["v"] += float(r["Counter_Value"])
    for did, e in per.items():
        tr = trace.get(str(did))
        e["us"] = (int(tr["End_Timestamp"]) - int(tr["Start_Timestamp"])) / 1e3 if tr else 0.0
    return per
fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
# same command, same launch sequence: dispatches pair up by order within each kernel name
def by_name(per):
    d = collections.defaultdict(list)
    for did in sorted(per): d[per[did]["name"]].append(per[did])
    return d
F, Wr = by_name(fetch), by_name(write)
print(open("$out/bench_accumulate_k32.txt").read().strip())
print()
print("HBM traffic per launch, 4096^2 D8 (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; B/cell = bytes / 16.8 M cells), kernel time from the FETCH pass")
print("%-44s %7s %9s %9s %9s %8s %7s" % ("kernel", "calls", "avg us", "B/cell", "GB/s", "% 8TB/s", "ms/real"))
k_real = 8 + 1   # realisations the profiled command makes (warm-up + K)
for name in F:
    if name not in Wr or len(F[name]) != len(Wr[name]): continue
    n = len(F[name])
    us = sum(e["us"] for e in F[name]) / n
    by = sum(2 * f["v"] + w["v"] for f, w in zip(F[name], Wr[name])) * 1024.0 / n
    if us <= 0: continue
    print("%-44s %7d %9.1f %9.2f %9.0f %8.1f %7.3f" % (name[:44], n, us, by / cells, by / us / 1e3, by / us / 1e3 / 80.0, n * us / 1e3 / k_real))
# rake-compress by round (26 launches per accumulation; the last accumulation of the run)
name = next((k for k in F if "k_rake_compress" in k), None)
if name:
    r = 26
    fs, ws = F[name][-r:], Wr[name][-r:]
    print()
    print("k_rake_compress, one accumulation, by round (SURVEY 8d bound: <= 144 B/cell/round streaming)")
    tot_b = tot_us = 0
    for i, (f, w) in enumerate(zip(fs, ws)):
        by = (2 * f["v"] + w["v"]) * 1024.0
        tot_b += by; tot_us += f["us"]
        print("  round %2d %8.1f us  fetch %8.2f  write %8.2f  B/cell  %7.0f GB/s" % (
            i, f["us"], 2 * f["v"] * 1024 / cells, w["v"] * 1024 / cells, by / max(f["us"], 1e-9) / 1e3))
    print("  all 26    %8.1f us  %8.2f B/cell in total" % (tot_us, tot_b / cells))
PY
cat $out/bench_accumulate.txt
du -sh $out
