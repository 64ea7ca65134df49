#!/bin/bash
# Kernel timeline of the last step of a bench run in stream order, with the queue each kernel ran on
# (the overlapped step: two particle chains side by side).   gpurun -- 'tools/trace_step.sh <out> [env...]'
out=/root/repo/${1:-gpurun_out/trace_step}; shift
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
env SOIL_BENCH_NO_EXACT=1 "$@" rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python /root/repo/bench.py ${BENCH_ARGS:---steps 2 --warmup 2} --no-cpu-baseline > /dev/null 2>&1
cd /root/repo
python - <<PY > $out/step.txt
import csv, glob, re
f = glob.glob("$out/t/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_erode_cells_fused" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
t0 = int(rows[lo]["Start_Timestamp"])
qs = {}
for r in rows[lo:ends[-1] + 1]:
    n = r["Kernel_Name"]
    m = re.search(r"k_tiled_round<(\d), \d, (\d+), \d+, (\d+), \w+, (\w+)", n)
    if m:
        name = "round %s %s" % (("F", "D")[int(m.group(1))], "sparse" if m.group(4) in ("true", "1") else "DENSE")
    else:
        name = re.sub(r"\(.*", "", n).replace("void soil::", "").replace("soil::", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    print("%9.1f .. %9.1f us (%7.1f)  q%d  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
PY
grep -v "k_fold\|k_pair_gate\|fillBuffer" $out/step.txt | head -${TRACE_LINES:-150}
