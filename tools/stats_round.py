"""What the deposits of the dense round kernel meet (lost swaps, lanes of a wave on one cell): runs
bench-like steps against the diagnostics build (libsoil_hip_stats.so, -DSOIL_STATS) and prints the
counters per launch kind.  Build in the container first:
    python -c "from soillib_amd import build; build.build(variant='stats', extra_flags=['-DSOIL_STATS'])"
then on the box:  SOIL_LIB=soillib_amd/lib/libsoil_hip_stats.so python tools/stats_round.py [size] [warm steps]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from soillib_amd import _abi, silt, soil  # noqa: E402
from soillib_amd.erosion import ErosionModel  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _abi.lib()
param = bench.script_param(soil)
model = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), param, S * S // 8, seed=0)
npar = soil.noise_t()
npar.seed = 3.0
npar.ext = [S, S]
bed = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
bench._interleave(lib, _abi, model.layers, bed)
silt.set(model.rainfall, 1.0)
silt.set(model.uplift, 0.0)
out = (C.c_ulonglong * 48)()
for step in range(WARM + 2):
    model.seed_step()
    model.particles_fluvial()
    model.particles_debris()
    model.cells_fused()
    model.swap_layers()
    model.step_index += 1
    _abi.check(lib.soil_device_synchronize())
    assert lib.soil_stats_read(out, 1) == 0
    if step < WARM:
        continue
    for kind in (0, 1):
        v = [out[kind * 24 + i] for i in range(24)]
        if v[0] == 0:
            continue
        it, lanes, dep, fol, lit, los, los_sh, dep_sh, dist, one = v[:10]
        print("step %d %s (dense round kernels): %d wave-iterations, %.1f lanes stepping, %.1f depositing (%.2f followers)"
              % (step, ("fluvial", "debris")[kind], it, lanes / it, dep / it, fol / it))
        print("   wave-iterations with a lost swap: %.1f %%; losers %.2f per such iteration (%.2f %% of the deposits), "
              "%.1f %% of them share the cell with another depositing lane of the wave"
              % (100.0 * lit / it, los / max(lit, 1), 100.0 * los / max(dep, 1), 100.0 * los_sh / max(los, 1)))
        print("   depositing lanes that share their cell with another lane of the wave: %.2f %%; distinct cells per "
              "lost iteration %.2f, lost iterations with one cell %.1f %%"
              % (100.0 * dep_sh / max(dep, 1), dist / max(lit, 1), 100.0 * one / max(lit, 1)))
        h = v[12:18]
        tot = max(sum(h), 1)
        print("   lanes with a lower lane of the wave on their cell: %.2f per iteration; nearest such lane at distance "
              "1: %.1f %%, 2: %.1f %%, 3: %.1f %%, 4-7: %.1f %%, 8-15: %.1f %%, 16+: %.1f %%; in the aligned pair %.1f %%, quad %.1f %%"
              % (sum(h) / it, *[100.0 * x / tot for x in h], 100.0 * v[18] / tot, 100.0 * v[19] / tot))
        print("   steps of length zero (a walker that stays where it is): %.2f %% of the steps; wave-iterations in which nobody else steps: "
              "%.2f %%, in which at most 8 others do: %.2f %%; wave-iterations of at most 8 stepping lanes: %.2f %%"
              % (100.0 * v[10] / max(lanes, 1), 100.0 * v[11] / it, 100.0 * v[23] / it, 100.0 * v[3] / it))
        print("   star of the iteration before as the prediction: %.2f followers per iteration; lanes that still share a cell "
              "among the swappers %.2f per iteration, iterations with any %.1f %%"
              % (v[20] / it, v[21] / it, 100.0 * v[22] / it))
