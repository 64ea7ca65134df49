"""Where a wave of k_tiled_round spends its time: runs bench-like steps against the
diagnostics build (libsoil_hip_prof.so, -DSOIL_PROF: s_memtime stamps at the seams of an
iteration) and prints the per-segment share.  Build in the container first:
    python -c "from soillib_amd import build; build.build(variant='prof', extra_flags=['-DSOIL_PROF'])"
then on the box:  SOIL_LIB=soillib_amd/lib/libsoil_hip_prof.so python tools/prof_round.py [size]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from soillib_amd import _abi, silt, soil  # noqa: E402
from soillib_amd.erosion import ErosionModel  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 1       # steps before the profiled ones
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
out = (C.c_ulonglong * 32)()
names = ["stops sorted out", "refill", "head", "gather+deposit begin", "advance", "deposit finish",
         "survivors out", "barrier wait", "prologue", "flux flush"]
for step in range(WARM + 2):
    model.seed_step()
    model.particles_fluvial()
    model.particles_debris()
    model.cells_fused()
    model.swap_layers()
    model.step_index += 1
    _abi.check(lib.soil_device_synchronize())
    assert lib.soil_prof_read(out, 1) == 0
    if step < WARM:
        continue
    for kind in (0, 1):
        v = [out[kind * 16 + i] for i in range(16)]
        tot = sum(v[:10])
        if tot == 0:   # SOIL_TILED_VERBOSE reads (and clears) the counters round by round: see stderr
            continue
        print("step %d %s: %d wave-iterations in %d waves, %.0f ticks per wave-iteration (all segments)"
              % (step, ("fluvial", "debris")[kind], v[10], v[11], tot / max(v[10], 1)))
        for i in range(10):
            print("   %-22s %5.1f %%  %7.0f ticks/iteration" % (names[i], 100.0 * v[i] / tot, v[i] / max(v[10], 1)))

prep = (C.c_ulonglong * 8)()
if lib.soil_prof_read_prepare(prep, 1) == 0:
    tot = sum(prep[:5])
    print("k_queue_prepare, thread 0, cycles summed over all launches: counts loaded %d | scans %d | start+order+jobs %d | "
          "cut jobs %d | published %d  (shares %s)" % (prep[0], prep[1], prep[2], prep[3], prep[4],
          " ".join("%.0f%%" % (100.0 * prep[i] / max(tot, 1)) for i in range(5))))
