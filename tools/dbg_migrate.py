"""Step counts of a world of slabs in migrate mode against the single-domain launches, kind by kind."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from util import script_param, to_gpu, to_np
from soillib_amd import silt, soil
from soillib_amd.erosion import ErosionModel
from soillib_amd.parallel import CallbackComm, SlabRunner
import test_gpu_parallel as T

world, S, W, maxage = [int(v) for v in (sys.argv[1:5] + [2, 64, 128, 48][len(sys.argv) - 1:])]
pp = script_param(soil.param_t()); pp.maxage = maxage
H = world * S
m = ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), pp, H * W // 8, seed=0)
q = soil.noise_t(); q.seed = 3.0; q.ext = [H, W]
bed = soil.noise(silt.shape(H, W), q, host=silt.gpu)
l0 = np.zeros((H, W, 2), np.float32); l0[..., 0] = to_np(bed)
m.set_layers(to_gpu(l0)); silt.set(m.rainfall, 1.0)
m.seed_step(); soil.particle_steps(reset=True)
m.particles_fluvial(); f = soil.particle_steps(reset=True)
m.particles_debris(); d = soil.particle_steps(reset=True)
print("single: fluvial", f, "debris", d, "water flux sum %.6e" % np.nansum(to_np(m.waterFlux), dtype=np.float64))
info = [None] * world
soil.particle_steps(reset=True)
got = T._run_world(world, S, W, pp, 1, maxage, info=info, mode=os.environ.get("MODE", "migrate"))
print("world:", soil.particle_steps(reset=True), [i["migration"] for i in info])
