mkdir -p gpurun_out/r3c
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r3c/pytest.log
for i in 1 2; do
for w in 0 2048 64; do
  SOIL_CELLS_WARM=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('warm $w', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms'].items()})" >> gpurun_out/r3c/ab.log
done
SOIL_PACK_PAIR=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pack separate', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms'].items()})" >> gpurun_out/r3c/ab.log
done
cat gpurun_out/r3c/pytest.log gpurun_out/r3c/ab.log
