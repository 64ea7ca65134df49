#!/bin/bash
# Duration of every kernel of ONE accumulation (4096^2 D8, the last of the run), in launch order:
#   gpurun -- 'tools/trace_rake_rounds.sh' -> gpurun_out/r06_rake/rounds_trace.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/trr
rocprofv3 --kernel-trace --output-format csv -d /tmp/trr -o p -- python /root/repo/tools/bench_accumulate.py --k 4 > /dev/null 2>&1
mkdir -p /root/repo/gpurun_out/r06_rake
python - <<'PY' | tee /root/repo/gpurun_out/r06_rake/rounds_trace${SUFFIX}.txt
import csv, glob, os
t = glob.glob('/tmp/trr/**/p_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
# the last k_donors launch starts the last accumulation
idx = max(i for i, r in enumerate(rows) if 'k_donors' in r['Kernel_Name'])
print("SOIL_RAKE_LIST_FROM=%s" % os.environ.get("SOIL_RAKE_LIST_FROM", "(default)"))
t0 = int(rows[idx]['Start_Timestamp'])
tot = 0.0
for r in rows[idx:]:
    n = r['Kernel_Name']
    if not any(k in n for k in ('k_donors', 'k_rake', 'k_mean')):
        break
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += us
    print("%9.1f us at %9.1f  %s" % (us, (int(r['Start_Timestamp']) - t0) / 1e3, n[:70]))
print("sum of kernel times %.1f us" % tot)
PY
