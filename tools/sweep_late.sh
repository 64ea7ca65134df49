run() { python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms'].items()})"; }
echo base; run
for sh in 8 9 10; do for sw in 2 3 4 6; do echo "debris late $sh from round $sw"; SOIL_TILED_LATE_D=$sh SOIL_TILED_SWITCH_D=$sw run; done; done
for sh in 8 9 10; do for sw in 9 10 11; do echo "fluvial late $sh from round $sw"; SOIL_TILED_LATE_F=$sh SOIL_TILED_SWITCH_F=$sw run; done; done
