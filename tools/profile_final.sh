#!/bin/bash
# Collects everything profiles/<name>/ holds, on the GPU box:
#   gpurun --timeout 1500 -- 'tools/profile_final.sh r01_final'
# then, back in the container:  python tools/profile_post.py r01_final
name=${1:-r01_final}
out=/root/repo/gpurun_out/$name; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
python /root/repo/bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o p -- python /root/repo/bench.py --steps $3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; }
pmc fetch "FETCH_SIZE" 2
pmc write "WRITE_SIZE" 2
pmc valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32" 1
pmc lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH" 1
python /root/repo/bench.py --size 1024 --steps 10000 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_c2_1024x10000.json
find $out -name "*.csv" | head -30; du -sh $out
