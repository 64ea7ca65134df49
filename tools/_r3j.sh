run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for i in 1 2; do
unset SOIL_TILED_TAILSCAN SOIL_PAIR_MODE SOIL_SCAN_LDS_PAD
LABEL="8192 default(serial,tail)" run --steps 10 --warmup 3
export SOIL_TILED_TAILSCAN=2 SOIL_PAIR_MODE=1
for pad in 0 40 80 92; do
export SOIL_SCAN_LDS_PAD=$pad
LABEL="8192 overlap standalone pad=$pad" run --steps 10 --warmup 3
done
export SOIL_SCAN_LDS_PAD=92
LABEL="4096 overlap standalone pad=92" run --size 4096 --steps 60 --warmup 5
unset SOIL_TILED_TAILSCAN SOIL_PAIR_MODE SOIL_SCAN_LDS_PAD
LABEL="4096 default" run --size 4096 --steps 60 --warmup 5
done
