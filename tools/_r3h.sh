(timeout 1500 python -m pytest tests -m gpu -x -v 2>&1 | grep -v "^  File\|^Extension\|PASSED" | tail -25)
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for pr in 1 2; do
export SOIL_PAIR_PRIORITY=$pr
LABEL="1024 prio=$pr" run --size 1024 --steps 3000 --warmup 50
LABEL="2048 prio=$pr" run --size 2048 --steps 300 --warmup 20
LABEL="4096 prio=$pr" run --size 4096 --steps 60 --warmup 5
LABEL="8192 prio=$pr" run --steps 10 --warmup 3
done
