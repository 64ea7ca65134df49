#!/usr/bin/env python
"""One 8192^2 step through ErosionModel.step() (fused cell kernel) and through the chain of
stand-alone operators a user of the reference's live API composes (step_unfused)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from soillib_amd import _abi, silt, soil
from soillib_amd.erosion import ErosionModel
from util import script_param
S=8192
param=script_param(soil.param_t()); param.maxage=256
m=ErosionModel(S,S,(20.0/S,20.0/S,4.0),param,S*S//8,seed=0)
p=soil.noise_t(); p.seed=3.0; p.ext=[S,S]
bed=soil.noise(silt.shape(S,S),p,host=silt.gpu)
z=silt.tensor(silt.float32,silt.shape(S,S),silt.gpu); silt.set(z,0.0)
_abi.check(_abi.lib().soil_layers_from_planes(m.layers.c_ptr,bed.c_ptr,z.c_ptr,S*S,None))
silt.set(m.rainfall,1.0)
for f,name in ((m.step,"fused step"),(m.step_unfused,"live-API op chain (unfused)")):
    f(); f()
    _abi.check(_abi.lib().soil_device_synchronize()); t0=time.perf_counter()
    for _ in range(5): f()
    _abi.check(_abi.lib().soil_device_synchronize()); print(name, "%.2f ms/step"%((time.perf_counter()-t0)/5*1e3))
