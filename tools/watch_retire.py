import sys, os
sys.path.insert(0, os.getcwd())
import bench
from soillib_amd import _abi, silt, soil
from soillib_amd.erosion import ErosionModel
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _abi.lib()
soil.debris_retire("watch")
param = bench.script_param(soil)
model = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), param, S * S // 8, seed=0)
npar = soil.noise_t(); npar.seed = 3.0; npar.ext = [S, S]
bed = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
bench._interleave(lib, _abi, model.layers, bed)
silt.set(model.rainfall, 1.0); silt.set(model.uplift, 0.0)
soil.particle_steps(reset=True)
for i in range(4):
    model.step()
    print("step", i, "steps walked", soil.particle_steps(reset=True), "violations", soil.debris_retire_violations(reset=False))
