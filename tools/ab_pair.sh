#!/bin/bash
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('serial', round(d['ms_per_step'],2), d['phases_ms'])"
for delay in 6 8 9 10 11; do
SOIL_PAIR_DELAY=$delay python bench.py --no-cpu-baseline --overlap-particles 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pair delay $delay', round(d['ms_per_step'],2), d['phases_ms'])"
done
done
