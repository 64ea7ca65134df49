run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for dl in 1 2 3; do for a in 1 2; do
  export SOIL_TILED_AHEAD=$a SOIL_PAIR_DELAY=$dl
  LABEL="1024 ahead=$a delay=$dl" run --size 1024 --steps 3000 --warmup 50
  LABEL="2048 ahead=$a delay=$dl" run --size 2048 --steps 300 --warmup 20
  LABEL="4096 ahead=$a delay=$dl" run --size 4096 --steps 60 --warmup 5
done; done
export SOIL_TILED_AHEAD=2 SOIL_PAIR_DELAY=2
LABEL="8192 ahead=2 delay=2" run --steps 10 --warmup 3
