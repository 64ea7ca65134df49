#!/bin/bash
# Where the waves of the round kernel spend their cycles (SQ wait / active counters, two passes).
#   gpurun -- 'tools/pmc_waves.sh gpurun_out/pmcw'
out=/root/repo/${1:-gpurun_out/pmcw}; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/a -o p -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_INSTS_VSKIPPED GRBM_GUI_ACTIVE --output-format csv -d $out/b -o p -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,collections
for sub in ('a','b'):
    rows=list(csv.DictReader(open('$out/%s/p_counter_collection.csv'%sub)))
    d=collections.OrderedDict()
    for r in rows:
        if 'k_tiled_round' in r['Kernel_Name']:
            key=(r['Dispatch_Id'], 'fluvial' if '<0,' in r['Kernel_Name'].replace(' ','')[:40] or 'ILi0' in r['Kernel_Name'] else 'debris')
            e=d.setdefault(key,{'dur_us':(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3})
            e[r['Counter_Name']]=e.get(r['Counter_Name'],0)+float(r['Counter_Value'])
    items=list(d.items())
    items=items[len(items)//2:]          # the second step
    for (did,kind),e in items[:6]+items[-14:-8]:
        w=e.get('SQ_WAVE_CYCLES',0)
        if w:
            print(kind, did, '%.0f us'%e['dur_us'], ' '.join('%s=%.3f'%(k.replace('SQ_',''),v/w) for k,v in e.items() if k not in ('dur_us','SQ_WAVE_CYCLES')))
        else:
            print(kind, did, '%.0f us'%e['dur_us'], ' '.join('%s=%.4g'%(k.replace('SQ_',''),v) for k,v in e.items() if k!='dur_us'))
PY
