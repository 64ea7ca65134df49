#!/bin/bash
# Timing experiments with parts of the round kernel switched off (diagnostics build, see
# erosion_particles_tiled.hip SOIL_ABLATE):  gpurun -- 'tools/ablate.sh [bench args]'
export SOIL_LIB=soillib_amd/lib/libsoil_hip_abl.so
for m in 0 1 2 4 8 10 "0 SOIL_TILED_STEPS=16" "0 SOIL_TILED_STEPS=24" "0 SOIL_TILED_STEPS=48" "1 SOIL_TILED_STEPS=16"; do
  set -- $m
  mask=$1; extra=$2
  env SOIL_ABLATE=$mask $extra python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate $mask $extra', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()}, d['particle_steps_per_step'])"
done
