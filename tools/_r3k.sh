(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pair_modes or pair_equals or without_rezeroing" 2>&1 | tail -4)
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phases_ms'].items()})"; }
for i in 1 2; do for pm in 1 2 3; do
export SOIL_PAIR_MODE=$pm
LABEL="8192 pair=$pm" run --steps 10 --warmup 3
LABEL="4096 pair=$pm" run --size 4096 --steps 60 --warmup 5
done; done
