#!/bin/bash
# the fused cell kernel against the relative placement of its 28 planes (tools/bench_cells.py --skew)
for skew in 0 4096 65536 1048576 1114112 2162688 0; do
  echo "skew $skew: $(python tools/bench_cells.py --skew $skew --iters 40 2>&1 | tail -1)"
done
