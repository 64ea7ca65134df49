(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12)
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for a in 0 2 4; do
  export SOIL_TILED_AHEAD=$a
  LABEL="1024 ahead=$a" run --size 1024 --steps 3000 --warmup 50
  LABEL="2048 ahead=$a" run --size 2048 --steps 300 --warmup 20
  LABEL="8192 ahead=$a" run --steps 10 --warmup 3
done
