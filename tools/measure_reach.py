"""How deep do particles spawned in one half of a grid walk into the other half?  (What a slab's
ghost zone really has to cover, against the worst case sqrt(2) * maxage = 362 rows.)
    python tools/measure_reach.py [size] [warm steps]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from soillib_amd import _abi, silt, soil  # noqa: E402
from soillib_amd.erosion import ErosionModel  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = _abi.lib()
param = bench.script_param(soil)
m = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), param, S * S // 8, seed=0)
npar = soil.noise_t()
npar.seed = 3.0
npar.ext = [S, S]
bench._interleave(lib, _abi, m.layers, soil.noise(silt.shape(S, S), npar, host=silt.gpu))
silt.set(m.rainfall, 1.0)
for step in range(warm + 1):
    # this step's particles of the upper half only: where do their deposits end up?
    half = ErosionModel.__new__(ErosionModel)
    half.__dict__.update(m.__dict__)
    half.dom = _abi.Domain(S, S, 0, S, 0, S // 2)
    for t in (m.waterFlux, m.massFlux, m.velocityFlux, m.debrisFlux, m.debrisVelocityFlux):
        silt.set(t, 0.0)
    m.seed_step()
    half.particles_fluvial()
    half.particles_debris()
    wf = m.waterFlux.cpu().numpy()
    df = m.debrisFlux.cpu().numpy()
    rows_f = np.nonzero((wf[S // 2:] != 0).any(axis=1))[0]
    rows_d = np.nonzero((df[S // 2:] != 0).any(axis=1))[0]
    dens = (wf[S // 2:] != 0).mean(axis=1)
    deep = [int(np.searchsorted(-np.maximum.accumulate(dens[::-1])[::-1], -q)) for q in (1e-2, 1e-3, 1e-4)]
    print("step %d: fluvial reach %d rows, debris reach %d rows (bound %d); rows below the boundary "
          "where > 1%% / 0.1%% / 0.01%% of the cells got a deposit: %s" % (
              step, rows_f.max() + 1 if len(rows_f) else 0, rows_d.max() + 1 if len(rows_d) else 0,
              int(lib.soil_ghost_rows(param._ref())), deep), flush=True)
    for t in (m.waterFlux, m.massFlux, m.velocityFlux, m.debrisFlux, m.debrisVelocityFlux):
        silt.set(t, 0.0)
    m.step()
