#!/bin/bash
run() { python bench.py --no-cpu-baseline --size $1 --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 min=$SOIL_TILED_AGG_MIN groups=$SOIL_TILED_AGG_GROUPS', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"; }
for size in 1024 2048 8192; do
  for cfg in "1000 6" "56 4" "48 4" "40 6" "32 8" "24 6" "16 12" "8 16"; do
    set -- $cfg
    export SOIL_TILED_AGG_MIN=$1 SOIL_TILED_AGG_GROUPS=$2
    run $size
  done
done
