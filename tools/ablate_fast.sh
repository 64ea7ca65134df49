#!/bin/bash
# Round 5: the round kernel with parts switched off (-DSOIL_ABLATE build), in either arithmetic, on the
# step after three ordinary ones (SOIL_ABLATE_AFTER): one timed step each, three alternations.
#   gpurun -- 'tools/ablate_fast.sh'
export SOIL_LIB=soillib_amd/lib/libsoil_hip_abl.so SOIL_BENCH_NO_EXACT=1 SOIL_ABLATE_AFTER=3
for i in 1 2 3; do
for arith in fast exact; do
for m in 0 2 4 8 10; do
  env SOIL_ABLATE=$m python bench.py --no-cpu-baseline --steps 1 --warmup 3 --particle-arith $arith --sequential-particles 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$arith ablate $m', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()}, d['particle_steps_per_step'])"
done; done; done
