run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for i in 1 2 3; do
unset SOIL_LIB
LABEL="8192 seq split" run --steps 8 --warmup 3 --sequential-particles
export SOIL_LIB=$PWD/soillib_amd/lib/libsoil_hip_nosplit.so
LABEL="8192 seq nosplit" run --steps 8 --warmup 3 --sequential-particles
done
unset SOIL_LIB
LABEL="1024 split" run --size 1024 --steps 3000 --warmup 50
export SOIL_LIB=$PWD/soillib_amd/lib/libsoil_hip_nosplit.so
LABEL="1024 nosplit" run --size 1024 --steps 3000 --warmup 50
