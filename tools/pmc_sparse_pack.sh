#!/bin/bash
# Vector instructions and LDS activity of the sparse round kernels with the small tiles packed into
# waves (default) and one tile per wave (SOIL_TILED_SPARSE_PACK=2):  gpurun -- 'tools/pmc_sparse_pack.sh'
cd /tmp; export TMPDIR=/tmp
for p in 2 1; do
  rm -rf /tmp/pp$p
  SOIL_TILED_SPARSE_PACK=$p rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d /tmp/pp$p -o p -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pp$p/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_tiled_round" in n and "true>" in n.split("(")[0]:
        acc[("fluvial" if "k_tiled_round<0" in n else "debris", r["Counter_Name"])] += float(r["Counter_Value"])
print("SOIL_TILED_SPARSE_PACK=$p", {k: "%.3g" % v for k, v in sorted(acc.items())})
PY
done
