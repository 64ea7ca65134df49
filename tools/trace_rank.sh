#!/bin/bash
# Kernel timeline of one step of an interior rank of an emulated world (tools/bench_rank_of_world.py) —
# where the rank's fixed costs are.   gpurun -- 'tools/trace_rank.sh <out> [bench_rank_of_world args]'
out=/root/repo/${1:-gpurun_out/trace_rank}; shift
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python /root/repo/tools/bench_rank_of_world.py --steps 1 --warmup 2 "$@" > $out/log.txt 2>&1
cd /root/repo
python - <<PY > $out/step.txt
import csv, glob, re, collections
f = glob.glob("$out/t/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_erode_cells_fused" in r["Kernel_Name"]]
# the last step: from behind the cell kernels of the step before (several launches: bands + interior)
last = ends[-1]
k = last
while k - 1 in ends or (k > 0 and "k_erode_cells" in rows[k - 1]["Kernel_Name"]): k -= 1
prev_end = max(i for i in ends if i < k - 3) if any(i < k - 3 for i in ends) else 0
lo = prev_end + 1
t0 = int(rows[lo]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in rows[lo:last + 1]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void soil::", "").replace("soil::", "")
    n = re.sub(r"k_tiled_round<(\d), \d, \d+, \d+, (\d+),.*", r"k_tiled_round<\1 lanes \2>", n)[:48]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(n, [0, 0.0, (s - t0) / 1e3])
    a[0] += 1; a[1] += (e - s) / 1e3
print("step wall %.1f us" % ((int(rows[last]["End_Timestamp"]) - t0) / 1e3))
for n, (c, t, first) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%9.1f us  x%-3d first at %9.1f  %s" % (t, c, first, n))
PY
cat $out/step.txt | head -40; tail -2 $out/log.txt
