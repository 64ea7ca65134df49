out=/root/repo/gpurun_out/r02_final/mix; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU --output-format csv -d $out -o p -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU --output-format csv -d ${out}_calib -o p -- /root/repo/tools/microbench/valu_issue > /dev/null 2>&1
python - <<PY
import csv,collections
for sub,pat in (("$out","k_tiled_round"),("${out}_calib","k<")):
    rows=list(csv.DictReader(open(sub+"/p_counter_collection.csv")))
    d=collections.OrderedDict()
    for r in rows:
        if pat in r["Kernel_Name"]:
            key=(r["Dispatch_Id"], r["Kernel_Name"][:34])
            e=d.setdefault(key,{})
            e[r["Counter_Name"]]=e.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    items=[(k,e) for k,e in d.items() if e.get("SQ_INSTS_VALU",0)>1e8]
    for (did,name),e in items[:40:3]:
        n=e["SQ_INSTS_VALU"]
        print(name, did, "VALU %.3g"%n, " ".join("%s=%.3f"%(k.replace("SQ_INSTS_VALU_",""),v/n) for k,v in e.items() if k!="SQ_INSTS_VALU"))
PY
