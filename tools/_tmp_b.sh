cd /tmp; export TMPDIR=/tmp
for ch in 1 2; do
export SOIL_BLUR_CHUNKS=$ch
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb$ch -o s -- python /root/repo/tools/bench_stencils.py 2>/dev/null | grep blur
python - <<PY
import csv,glob
f=glob.glob('/tmp/pb$ch/**/s_kernel_trace.csv',recursive=True)[0]
import collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'blur' in r['Kernel_Name']: d[(r['Kernel_Name'][:40],r['Grid_Size_X'],r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print($ch,k,len(v),'min %.1f med %.1f'%(min(v),sorted(v)[len(v)//2]))
PY
done
