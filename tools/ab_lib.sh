#!/bin/bash
# A/B of two builds of libsoil_hip.so on one box through the product path (overlapped launches):
#   tools/ab_lib.sh <base.so> [bench args]
base=$1; shift
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export SOIL_LIB=$base; else unset SOIL_LIB; fi
    SOIL_PAIR_DELAY=2 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"
  done
done
