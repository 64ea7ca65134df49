#!/usr/bin/env python
"""The live binding of the reference always carries the colour (albedo) planes through the
transport calls (python/source/model.cpp:237-295).  Times soil.transport_fluvial /
transport_debris with them on an S^2 grid, N = cells/8."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from soillib_amd import _abi, silt, soil  # noqa: E402
from util import script_param  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", type=int, default=0)
args = ap.parse_args()
S = args.size
lib = _abi.lib()
lib.soil_set_particle_mode(args.mode)
param = script_param(soil.param_t())
param.maxage = 256
scale = (20.0 / S, 20.0 / S, 4.0)
p = soil.noise_t()
p.seed = 3.0
p.ext = [S, S]
bed = soil.noise(silt.shape(S, S), p, host=silt.gpu)


def z(*dims):
    t = silt.tensor(silt.float32, silt.shape(*dims), silt.gpu)
    silt.set(t, 0.0)
    return t


layers = z(S, S, 2)
_abi.check(lib.soil_layers_from_planes(layers.c_ptr, bed.c_ptr, z(S, S).c_ptr, S * S, None))
rain = z(S, S)
silt.set(rain, 1.0)
wh, wf, m, mf, v, vf = z(S, S), z(S, S), z(S, S), z(S, S), z(S, S, 2), z(S, S, 2)
d, df, dv, dvf = z(S, S), z(S, S), z(S, S, 2), z(S, S, 2)
ab, af, adf, asrc = z(S, S, 3), z(S, S, 3), z(S, S, 3), z(S, S, 3)
silt.set(asrc, 0.5)
N = S * S // 8
rng = silt.tensor(silt.rng, silt.shape(N), silt.gpu)
for colour in (False, True):
    times = []
    for rep in range(args.reps + 1):
        silt.seed(rng, 0, rep * N)
        for t in (wf, mf, vf, df, dvf, af, adf):
            silt.set(t, 0.0)
        _abi.check(lib.soil_device_synchronize())
        t0 = time.perf_counter()
        soil.transport_fluvial(layers, rain, wh, wf, m, mf, v, vf, ab if colour else None,
                               af if colour else None, asrc if colour else None, rng, scale, param)
        _abi.check(lib.soil_device_synchronize())
        t1 = time.perf_counter()
        soil.transport_debris(layers, dv, dvf, d, df, ab if colour else None,
                              adf if colour else None, asrc if colour else None, rng, scale, param)
        _abi.check(lib.soil_device_synchronize())
        t2 = time.perf_counter()
        if rep:
            times.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    f = sum(t[0] for t in times) / len(times)
    g = sum(t[1] for t in times) / len(times)
    print("%dx%d %s colour planes: transport_fluvial %.1f ms, transport_debris %.1f ms" % (
        S, S, "with" if colour else "without", f, g))
