#!/bin/bash
# one-box sweep of the tiled-transport knobs: tools/sweep.sh
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()})"; }
run default
for k in 24 28 36 40 48; do SOIL_TILED_STEPS=$k run steps=$k; done
for m in 2000 3000 6000 8000; do SOIL_TILED_FINISH_MRATE=$m run mrate=$m; done
SOIL_TILED_SHAPE=0 run shape=0
SOIL_TILED_SHAPE=1 run shape=1
SOIL_TILED_SHAPE=2 run shape=2
