#!/bin/bash
# Round-6 diagnosis of the round-5 hang: the C++ test binaries bare (no Python, no
# torch in the process), each RCCL stage alone with NCCL_DEBUG=INFO, then all stages in one process.
# Usage (GPU box): bash tools/diag_rccl.sh   -> gpurun_out/rccl/*.log
set -u
O=gpurun_out/rccl
mkdir -p $O
L=$PWD/soillib_amd/lib
for n in test_cpp_api test_cpp_rccl; do
  g++ -std=c++17 -O1 -pthread -I include tests/cpp/$n.cpp -o /tmp/$n -L $L -lsoil_hip -Wl,-rpath,$L -Wl,-rpath,/opt/rocm/lib || exit 9
done
{
  ls -la /opt/rocm/lib/librccl* 2>&1
  python - <<'PY'
import glob, os, torch
print("torch", torch.__version__, torch.__file__)
print(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl*")))
PY
  ldd /tmp/test_cpp_rccl | grep -i "hip\|rccl\|hsa"
  env | grep -i "nccl\|rccl\|hsa\|hip\|rocr\|LD_" | sort
} > $O/env.log 2>&1
t0=$(date +%s.%N)
timeout 100 /tmp/test_cpp_api > $O/api.log 2>&1; echo "api rc=$? $(echo "$(date +%s.%N) - $t0" | bc) s" | tee -a $O/summary.log
for stage in slab self group reduce; do
  t0=$(date +%s.%N)
  NCCL_DEBUG=INFO timeout 100 /tmp/test_cpp_rccl $stage > $O/$stage.log 2>&1
  echo "$stage rc=$? $(echo "$(date +%s.%N) - $t0" | bc) s" | tee -a $O/summary.log
done
t0=$(date +%s.%N)
NCCL_DEBUG=INFO timeout 120 /tmp/test_cpp_rccl > $O/all.log 2>&1; echo "all rc=$? $(echo "$(date +%s.%N) - $t0" | bc) s" | tee -a $O/summary.log
t0=$(date +%s.%N)
timeout 120 /tmp/test_cpp_rccl > $O/all_quiet.log 2>&1; echo "all_quiet rc=$? $(echo "$(date +%s.%N) - $t0" | bc) s" | tee -a $O/summary.log
grep -h "MARK\|WATCHDOG\|RCCL_\|FAIL\|OK" $O/all_quiet.log | tail -40
