(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5)
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for i in 1 2; do
LABEL="8192 seq" run --steps 10 --warmup 3 --sequential-particles
LABEL="8192" run --steps 10 --warmup 3
LABEL="1024" run --size 1024 --steps 3000 --warmup 50
done
