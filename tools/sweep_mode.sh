#!/bin/bash
for size in 640 768 896; do
  for mode in 2 3; do
    python bench.py --no-cpu-baseline --size $size --steps 10 --warmup 3 --particle-mode $mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$size mode $mode', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"
  done
done
