#!/bin/bash
# Where a 1024^2 step goes (BASELINE configs[1]): per-round queue statistics and a kernel trace.
#   gpurun -- 'tools/prof_c2.sh <outdir>'
out=${1:-gpurun_out/c2}; mkdir -p $out
SOIL_TILED_VERBOSE=1 python bench.py --size 1024 --steps 2 --warmup 60 --no-cpu-baseline > $out/verbose.json 2> $out/verbose.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/trace -o t -- python /root/repo/bench.py --size 1024 --steps 20 --warmup 60 --no-cpu-baseline > /root/repo/$out/bench_traced.json 2>/dev/null
cd /root/repo
python - <<PY
import csv,glob,collections
f=glob.glob('$out/trace/**/t_kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:12]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), '%8.1f us avg'%(float(r['AverageNs'])/1e3), '%7.3f ms/step'%(int(r['TotalDurationNs'])/1e6/80), r['Percentage'])
PY
