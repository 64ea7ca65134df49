#!/bin/bash
# A/B of two builds at several grid sizes: tools/ab_sizes.sh <base.so>
base=$1
for size in 1024 2048 4096 8192; do
  for which in base new base new; do
    if [ $which = base ]; then export SOIL_LIB=$base; else unset SOIL_LIB; fi
    python bench.py --no-cpu-baseline --size $size --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$size $which', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"
  done
done
