#!/bin/bash
# Vector / scalar / memory instructions of the k_rake_compress launches of one accumulation (4096^2 D8):
#   gpurun -- 'tools/pmc_rake.sh'
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pr
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE} --output-format csv -d /tmp/pr -o p -- python /root/repo/tools/bench_accumulate.py --k 4 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pr/**/p_counter_collection.csv',recursive=True)[0]
t=glob.glob('/tmp/pr/**/p_kernel_trace.csv',recursive=True)[0]
tr={r['Dispatch_Id']:r for r in csv.DictReader(open(t))}
per=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_rake' not in r['Kernel_Name']: continue
    e=per.setdefault(r['Dispatch_Id'],{})
    e[r['Counter_Name']]=e.get(r['Counter_Name'],0)+float(r['Counter_Value'])
ids=list(per)[-26:]
import os
if os.environ.get('PMC'):
    names=sorted({k for d in ids for k in per[d]})
    print('round      us  '+'  '.join('%16s'%n for n in names)+'   (per cell)')
    for i,d in enumerate(ids):
        e=per[d]; us=(int(tr[d]['End_Timestamp'])-int(tr[d]['Start_Timestamp']))/1e3
        print('%3d %8.1f  '%(i,us)+'  '.join('%16.3f'%(e.get(n,0)/(4096*4096.0)) for n in names))
    raise SystemExit

print('round    us   VALU/cell SALU/cell VMEM_RD/cell VMEM_WR/cell  waves  VALU x 4 cycles / (1024 SIMDs x us x 2.4e3)')
for i,d in enumerate(ids):
    e=per[d]; us=(int(tr[d]['End_Timestamp'])-int(tr[d]['Start_Timestamp']))/1e3
    cells=4096*4096/64.0
    print('%3d %8.1f %9.1f %9.1f %9.2f %9.2f %8d   %.2f'%(i,us,e['SQ_INSTS_VALU']/cells,e['SQ_INSTS_SALU']/cells,e['SQ_INSTS_VMEM_RD']/cells,e['SQ_INSTS_VMEM_WR']/cells,e['SQ_WAVES'],e['SQ_INSTS_VALU']*4/(1024*us*2.4e3)))
PY
