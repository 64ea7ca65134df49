#!/bin/bash
# Several builds of libsoil_hip.so alternated on one box through the product path:
#   tools/ab_libs.sh "<bench args>" <lib.so|default> ...
args=$1; shift
for i in 1 2 3; do
  for l in "$@"; do
    if [ $l = default ]; then unset SOIL_LIB; else export SOIL_LIB=$l; fi
    python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s]' % '$l'.split('/')[-1], round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"
  done
done
