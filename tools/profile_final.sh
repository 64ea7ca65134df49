#!/bin/bash
# Collects everything profiles/<name>/ holds, on the GPU box:
#   gpurun --timeout 2400 -- 'tools/profile_final.sh r04_final'
# then, back in the container:  python tools/profile_post.py r04_final
name=${1:-r06_final}
out=/root/repo/gpurun_out/$name; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
# the line as the driver runs it: exact (reference) particle arithmetic, the fast mode timed beside it (`fast_arithmetic`)
python /root/repo/bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/bench_line.json
python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --particle-arith fast 2>/dev/null | tail -1 > $out/bench_line_fast.json
export SOIL_BENCH_NO_OTHER_ARITH=1   # the profiled passes below: one arithmetic (exact, the default) per process
# the same step with the two particle launches one after the other: per-launch phase timings, and
# kernel durations / counters that are not mixed with the other launch's kernels
python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sequential-particles 2>/dev/null | tail -1 > $out/bench_line_sequential.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sequential-particles > /dev/null 2>&1
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o p -- python /root/repo/bench.py --steps $3 --warmup 1 --no-cpu-baseline --sequential-particles > /dev/null 2>&1; }
pmc fetch "FETCH_SIZE" 2
pmc write "WRITE_SIZE" 2
# the default step (both launches overlapped, flux planes left to the next step's first rounds): the
# traffic of the 84-byte flavour of the fused cell kernel and of the storing first round
pmcd() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o p -- python /root/repo/bench.py --steps $3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; }
pmcd fetch_lazy "FETCH_SIZE" 3
pmcd write_lazy "WRITE_SIZE" 3
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -o s -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
pmc valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32" 1
pmc lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH" 1
pmc mix "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE" 1
# what the VALU counters read on instruction streams of known cost (the opcode classes of the
# microbenchmark): the calibration of `valu_busy`
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/calib -o p -- /root/repo/tools/microbench/valu_issue > $out/valu_issue.txt 2>&1
# the graph / stencil / conditioning kernels (BASELINE configs[2] and SURVEY 8a rows a8-a13)
# API-level times un-profiled (under --kernel-trace they read ~10 % high), kernel durations from a profiled run
python /root/repo/tools/bench_stencils.py > $out/bench_stencils.txt 2>/dev/null
python /root/repo/tools/bench_accumulate.py > $out/bench_accumulate.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stencils -o s -- python /root/repo/tools/bench_stencils.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/accumulate -o s -- python /root/repo/tools/bench_accumulate.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/stencils_fetch -o p -- python /root/repo/tools/bench_stencils.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/stencils_write -o p -- python /root/repo/tools/bench_stencils.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/accumulate_fetch -o p -- python /root/repo/tools/bench_accumulate.py --k 8 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/accumulate_write -o p -- python /root/repo/tools/bench_accumulate.py --k 8 > /dev/null 2>&1
unset SOIL_BENCH_NO_OTHER_ARITH
# the other BASELINE configs in the driver's line format (bench.py --config)
for c in c1 c2 c3; do python /root/repo/bench.py --config $c 2>/dev/null | tail -1 > $out/bench_$c.json; done
# drop the raw per-dispatch tables of the big passes once summarised? (kept: they fit the 64 MiB budget)
find $out -name "*.csv" | head -40; du -sh $out
