#!/bin/bash
# A/B of two builds of libsoil_hip.so on one box, stencil set: tools/ab_stencils.sh <base.so> [pattern]
base=$1; pat=${2:-ms}
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export SOIL_LIB=$base; else unset SOIL_LIB; fi
    python tools/bench_stencils.py --reps 20 2>/dev/null | grep " ms " | grep -E "$pat" | sed "s/^/$which /" | cut -c1-52
  done
done
