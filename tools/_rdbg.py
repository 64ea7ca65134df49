import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import pyoracle as oracle
from util import product_param, rng_to_gpu, script_param, terrain, to_gpu, to_np
from soillib_amd import _abi, soil
import test_debris_retire as T
lib = _abi.lib()
H, W = 256, 192
op = oracle.default_param(); op.maxage = 128
pp = product_param(op)
N = H * W // 8
scale = (20.0 / H, 20.0 / W, 4.0)
layers = terrain(oracle, H, W, sediment=0.01)
vel0 = (np.random.default_rng(5).standard_normal((H, W, 2)) * 0.5).astype(np.float32)
lib.soil_set_particle_mode(3)
soil.debris_retire("watch")
print(T._debris_launch(soil, layers, vel0, N, scale, pp)[0], soil.debris_retire_violations())
out = (C.c_float * 16)()
print(lib.soil_retire_dbg_read(out), list(out))
for k in ["viscosityDebris","bedShearDebris","yieldStress","depositionRateDebris","suspensionRateDebris","landslideRateDebris","critSlopeBedrock","gravity"]:
    print(k, getattr(op,k))
