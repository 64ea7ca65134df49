#!/bin/bash
# Is the round-5 hang a race?  N bare runs of tests/cpp/test_cpp_rccl back to back, each under its own
# watchdog; prints the slowest run and every run that did not end with CPP_RCCL_OK.
set -u
N=${1:-30}
O=gpurun_out/rccl_loop
mkdir -p $O
L=$PWD/soillib_amd/lib
g++ -std=c++17 -O1 -pthread -I include tests/cpp/test_cpp_rccl.cpp -o /tmp/test_cpp_rccl -L $L -lsoil_hip -Wl,-rpath,$L -Wl,-rpath,/opt/rocm/lib || exit 9
bad=0
for i in $(seq 1 $N); do
  s=$(date +%s%N)
  SOIL_TEST_WATCHDOG_S=40 timeout 60 /tmp/test_cpp_rccl > $O/run_$i.log 2>&1
  rc=$?
  e=$(date +%s%N)
  ms=$(( (e - s) / 1000000 ))
  echo "run $i rc=$rc ${ms} ms $(grep -c CPP_RCCL_OK $O/run_$i.log)" >> $O/summary.txt
  if [ $rc -ne 0 ]; then bad=$((bad+1)); tail -5 $O/run_$i.log; else rm -f $O/run_$i.log; fi
done
sort -k4 -n -t' ' $O/summary.txt | tail -3
echo "bad runs: $bad of $N"
