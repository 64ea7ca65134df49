out=/root/repo/gpurun_out/pmc_lds_$1; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out -o p -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sequential-particles > /dev/null 2>&1
cd /root/repo
python - <<PY
import csv,glob,collections
f=glob.glob('$out/**/p_counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
per=collections.OrderedDict()
for r in rows:
    if 'k_tiled_round<0' not in r['Kernel_Name']: continue
    per.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value'])
ds=list(per.values()); ds=ds[len(ds)//2:]
for i,d in enumerate(ds):
    cyc=d['GRBM_GUI_ACTIVE']/8
    print(i,'lds instr %.1fM'%(d['SQ_INSTS_LDS']/1e6),'busy %.3f'%(d['SQ_LDS_IDX_ACTIVE']/(256*cyc)),'conflict %.3f'%(d['SQ_LDS_BANK_CONFLICT']/(256*cyc)),'cycles %.2fM'%(cyc/1e6))
PY
