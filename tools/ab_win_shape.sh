#!/bin/bash
# The window stencils with the rows landing in LDS (window.hpp: SOIL_WIN_SHAPE 0 band walk in registers, 1 band walk through LDS,
# 2 flat), alternated on one box: tools/ab_win_shape.sh [pattern]
pat=${1:-steepest|direction d8}
for i in 1 2 3; do
  for v in ${SHAPES:-0 1 2 3}; do
    SOIL_WIN_SHAPE=$v python tools/bench_stencils.py --reps 20 2>/dev/null | grep " ms " | grep -E "$pat" | sed "s/^/shape=$v /" | cut -c1-100
  done
done
