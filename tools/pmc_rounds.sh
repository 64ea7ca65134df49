#!/bin/bash
# PMC passes over one bench step (run on the GPU box): tools/pmc_rounds.sh <outdir>
out=$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o p -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; }
run a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32"
run b "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run c "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_WAVES"
ls -R $out | head -20
