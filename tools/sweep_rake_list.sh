#!/bin/bash
# Round 6: the round from which the rake-compress rounds run over the list of pending cells
# (SOIL_RAKE_LIST_FROM; 0 = dense rounds throughout, as in round 5).  bench.py --config c3, K = 64.
O=gpurun_out/r06_rake
mkdir -p $O
for f in ${FROMS:-0 1 2 3 4 5}; do
  SOIL_RAKE_LIST_FROM=$f python bench.py --config c3 --steps 64 --no-cpu-baseline > $O/c3_from$f.json 2> $O/c3_from$f.err
  python - "$f" "$O/c3_from$f.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("list_from %s: %.3f ms per realisation (device %.3f), one accumulate call %.3f ms, random_weighted %.3f ms" % (
    sys.argv[1], d["ms_per_step"], d["config"]["device_ms_per_realisation"], d["config"]["one_accumulate_call_ms"],
    d["config"]["one_random_weighted_call_ms"]))
PY
done | tee $O/sweep.txt
