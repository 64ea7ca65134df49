#!/bin/bash
# Round 6: what the lost-swap path costs.  -DSOIL_ABLATE build, exact arithmetic, launches one by one, the step
# after three ordinary ones; masks 0 (full) 16 (lost swaps dropped) 4 (plain stores) 2 (no deposits).
#   gpurun -- 'tools/ablate_r6.sh [size]'
size=${1:-8192}
export SOIL_LIB=soillib_amd/lib/libsoil_hip_abl.so SOIL_BENCH_NO_OTHER_ARITH=1 SOIL_ABLATE_AFTER=3 SOIL_TILED_BUDDY=0
for i in 1 2; do
for m in 0 16 48 4 2; do
for extra in "" ; do
  env SOIL_ABLATE=$m $extra python bench.py --no-cpu-baseline --size $size --steps 1 --warmup 3 --sequential-particles 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('exact ablate $m $extra', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()}, d['particle_steps_per_step'])"
done; done; done
for r in 0 1 2; do
  env SOIL_ABLATE=0 SOIL_TILED_RETRIES=$r python bench.py --no-cpu-baseline --size $size --steps 1 --warmup 3 --sequential-particles 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('exact retries $r', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()}, d['particle_steps_per_step'])"
done
