import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from soillib_amd import _abi, silt, soil
import test_gpu_fullsize as T
hip = _abi.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = T._model(S)
m.step(); m.step()
for rep in range(2):
    for mode in (1, 2, 3):
        hip.soil_set_particle_mode(mode)
        for t in (m.waterFlux, m.massFlux, m.velocityFlux, m.debrisFlux, m.debrisVelocityFlux):
            silt.set(t, 0.0)
        m.seed_step()
        soil.particle_steps(reset=True)
        m.particles_fluvial(); f = soil.particle_steps(reset=True)
        m.particles_debris(); d = soil.particle_steps(reset=True)
        print(S, 'mode', mode, 'fluvial', f, 'debris', d, flush=True)
