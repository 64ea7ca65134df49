"""Particle-step counts per launch kind, exact against fast arithmetic, over a few steps of one model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from util import script_param, to_gpu, to_np
from soillib_amd import silt, soil
from soillib_amd.erosion import ErosionModel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
param = script_param(soil.param_t())
for mode in ("exact", "fast"):
    soil.particle_arith(mode)
    m = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), param, S * S // 8, seed=1)
    q = soil.noise_t(); q.seed = 3.0; q.ext = [S, S]
    bed = soil.noise(silt.shape(S, S), q, host=silt.gpu)
    layers0 = np.zeros((S, S, 2), np.float32); layers0[..., 0] = to_np(bed)
    m.set_layers(to_gpu(layers0)); silt.set(m.rainfall, 1.0)
    for step in range(3):
        m.seed_step()
        soil.particle_steps(reset=True)
        m.particles_fluvial(); f = soil.particle_steps(reset=True)
        m.particles_debris(); d = soil.particle_steps(reset=True)
        m.cells_fused(); m.swap_layers(); m.step_index += 1
        print(mode, "step", step, "fluvial", f, "debris", d, "water sum %.6e" % np.nansum(to_np(m.waterHeight), dtype=np.float64),
              "debris sum %.6e" % np.nansum(to_np(m.debris), dtype=np.float64))
soil.particle_arith("exact")
