#!/bin/bash
# the fused cell kernel against where its 28 planes live (tools/bench_cells.py): separate
# allocations (what the library's pool does), one arena (--arena, --chunk-mb), padding
# between the planes (--skew); several process starts each
for rep in 1 2 3; do
  for opt in "" "--skew 1048576" "--arena" "--arena --skew 35651584" "--arena --chunk-mb 1024"; do
    echo "[$opt] $(python tools/bench_cells.py $opt --iters 40 2>&1 | tail -1)"
  done
done
