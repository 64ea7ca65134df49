#!/bin/bash
# Per-kernel averages of tools/bench_accumulate.py --k 8 under rocprofv3 --stats, by SOIL_WIN_SHAPE:
#   gpurun -- 'tools/acc_stats.sh "0 4 5"'
cd /tmp; export TMPDIR=/tmp
for shp in ${1:-0 4 5}; do
rm -rf /tmp/accs; SOIL_WIN_SHAPE=$shp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/accs -o s -- python /root/repo/tools/bench_accumulate.py --k 8 2>/dev/null | tail -1
python - <<PY
import csv,glob
f=glob.glob('/tmp/accs/**/s_kernel_stats.csv',recursive=True)[0]
print('SOIL_WIN_SHAPE=$shp')
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('k_donors4','k_rake','k_random','k_mean')):
        print('  %-60s calls %4s avg %8.1f us'%(n[:60],r['Calls'],float(r['AverageNs'])/1e3))
PY
done
