(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -8)
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for ts in 1 2; do for pm in 1 3; do
export SOIL_TILED_TAILSCAN=$ts SOIL_PAIR_MODE=$pm
LABEL="1024 tail=$ts pair=$pm" run --size 1024 --steps 3000 --warmup 50
LABEL="2048 tail=$ts pair=$pm" run --size 2048 --steps 300 --warmup 20
LABEL="4096 tail=$ts pair=$pm" run --size 4096 --steps 60 --warmup 5
LABEL="8192 tail=$ts pair=$pm" run --steps 10 --warmup 3
done; done
