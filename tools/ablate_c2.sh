#!/bin/bash
# The 1024^2 step with channels formed (60 warm-up steps), parts of the round kernel switched off
# for the 20 timed steps only.   gpurun -- 'tools/ablate_c2.sh'
export SOIL_LIB=soillib_amd/lib/libsoil_hip_abl.so SOIL_ABLATE_AFTER=60
for m in 0 2 4 8 "0 SOIL_TILED_STEPS=64" "0 SOIL_TILED_AGG_MIN=16" "0 SOIL_TILED_AGG_MIN=64" "0 SOIL_TILED_DEP=1" "0 SOIL_TILED_SHAPE=0" "0 SOIL_TILED_SHAPE=2"; do
  set -- $m
  mask=$1; extra=$2
  env SOIL_ABLATE=$mask $extra python bench.py --size 1024 --no-cpu-baseline --steps 20 --warmup 60 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate $mask $extra', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms'].items()}, d['particle_steps_per_step'])"
done
