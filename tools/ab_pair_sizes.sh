#!/bin/bash
# both particle launches of a step back to back vs overlapped (soil_particles_pair_slab), by grid size
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"; }
for spec in "1024 1000" "2048 100" "4096 20"; do set -- $spec
  echo "size $1 serial"; run --size $1 --steps $2 --warmup 5
  for delay in 1 2 4 8; do echo "size $1 pair, debris starts at fluvial round $delay"; SOIL_PAIR_DELAY=$delay run --size $1 --steps $2 --warmup 5 --overlap-particles; done
done
