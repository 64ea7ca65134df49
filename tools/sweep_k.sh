#!/bin/bash
run() { python bench.py --no-cpu-baseline --size $1 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 K=$SOIL_TILED_STEPS', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"; }
for size in 512 1024 2048 4096; do
  for k in 32 40 48 64 96; do
    export SOIL_TILED_STEPS=$k
    run $size
  done
done
