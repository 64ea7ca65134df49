#!/bin/bash
# Kernel timeline of the last step of a sequential-launch bench run: every kernel of the two particle
# launches in stream order with its duration (the round kernels tagged dense / sparse).
#   gpurun -- 'tools/trace_rounds.sh gpurun_out/r04_trace [extra env assignments]'
#   BENCH_ARGS="--size 1024 --steps 3 --warmup 300" picks another workload (default: the sequential 8192^2 step)
out=/root/repo/${1:-gpurun_out/r04_trace}; shift
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python /root/repo/bench.py ${BENCH_ARGS:---steps 2 --warmup 1 --sequential-particles} --no-cpu-baseline > /dev/null 2>&1
cd /root/repo
python - <<PY > $out/rounds.txt
import csv, glob, re
f = glob.glob("$out/t/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last k_tiled_pack / k_rng_seed before the final k_erode_cells_fused
ends = [i for i, r in enumerate(rows) if "k_erode_cells_fused" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:ends[-1] + 1]:
    n = r["Kernel_Name"]
    m = re.search(r"k_tiled_round<(\d), \d, (\d+), \d+, (\d+), \w+, (\w+)>", n)
    if m:
        name = "round %s %s (%s rows, %s lanes)" % (("fluvial", "debris")[int(m.group(1))], "SPARSE" if m.group(4) in ("true", "1") else "dense", m.group(2), m.group(3))
    else:
        name = re.sub(r"\(.*", "", n).replace("void soil::", "").replace("soil::", "")[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
PY
cat $out/rounds.txt | grep -v "k_fold" | head -${TRACE_LINES:-120}
