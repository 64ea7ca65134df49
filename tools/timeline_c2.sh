#!/bin/bash
# Timeline of one late step of a small grid: kernel start/end from a rocprofv3 kernel trace,
# busy time of the GPU (union of kernel intervals) against the step's wall time.
#   gpurun -- 'tools/timeline_c2.sh <outdir> [size]'
out=${1:-gpurun_out/tl}; size=${2:-1024}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
SOIL_BENCH_NO_EXACT=1 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$out/trace -o t -- python /root/repo/bench.py --size $size --steps 8 --warmup ${WARMUP:-60} --no-cpu-baseline > /root/repo/$out/bench_traced.json 2>/dev/null
cd /root/repo
python - <<PY
import csv,glob
f=glob.glob('$out/trace/**/t_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steps end with the fused cell kernel
ends=[i for i,r in enumerate(rows) if 'k_erode_cells_fused' in r['Kernel_Name']]
a,b=ends[-3]+1,ends[-2]+1
step=rows[a:b]
t0=int(step[0]['Start_Timestamp']); t1=int(step[-1]['End_Timestamp'])
busy=0; cur_s=cur_e=None
for r in step:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print('step wall %.1f us, GPU busy (union) %.1f us, kernels %d'%((t1-t0)/1e3,busy/1e3,len(step)))
import collections
agg=collections.defaultdict(lambda:[0,0])
for r in step:
    n=r['Kernel_Name'].split('(')[0].replace('void soil::','').replace('soil::','')[:60]
    agg[n][0]+=1; agg[n][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): print('%-62s %4d %9.1f us'%(n,c,t/1e3))
with open('$out/timeline.txt','w') as g:
    for r in step:
        g.write('%9.1f %8.1f %s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'].split('(')[0][:80]))
PY
