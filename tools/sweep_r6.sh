#!/bin/bash
# Round 6, after the debris retirement: the fluvial launch has the chip to itself — knobs re-swept.
#   gpurun -- 'tools/sweep_r6.sh'
export SOIL_BENCH_NO_OTHER_ARITH=1
tools/ab_env.sh "--steps 10 --warmup 2" "" "SOIL_TILED_STEPS_F=36" "SOIL_TILED_STEPS_F=40" "SOIL_TILED_STEPS_F=48" "SOIL_TILED_STEPS_F=56" "SOIL_TILED_STEPS_F=64"
tools/ab_env.sh "--steps 10 --warmup 2" "" "SOIL_PAIR_MODE=1" "SOIL_PAIR_MODE=3" "SOIL_STEP_PAIR=0" "SOIL_TILED_AHEAD=3" "SOIL_TILED_AHEAD=1"
tools/ab_env.sh "--steps 10 --warmup 2" "" "SOIL_TILED_TAIL=100000" "SOIL_TILED_TAIL=400000" "SOIL_TILED_SPARSE_PCT=10" "SOIL_TILED_SPARSE_PCT=50" "SOIL_TILED_FINISH_MRATE=8000"
