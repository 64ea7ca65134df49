run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
export SOIL_PAIR_MODE=2
for e in 0 128 256 512 1024; do
export SOIL_PAIR_EARLY=$e
LABEL="8192 early=$e" run --steps 10 --warmup 3
LABEL="4096 early=$e" run --size 4096 --steps 60 --warmup 5
done
unset SOIL_PAIR_EARLY
for pm in 1 2; do
export SOIL_PAIR_MODE=$pm
LABEL="2048 pair=$pm" run --size 2048 --steps 300 --warmup 20
LABEL="1024 pair=$pm" run --size 1024 --steps 3000 --warmup 50
done
