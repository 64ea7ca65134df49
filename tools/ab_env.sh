#!/bin/bash
# One build, several environments, alternated on one box through the product path (overlapped launches):
#   tools/ab_env.sh "<bench args>" "VAR=1 VAR2=3" "VAR=2" ...     ("" = defaults)
args=$1; shift
for i in 1 2 3; do
  for e in "$@"; do
    env $e python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s]' % '$e', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms'].items()})"
  done
done
