#!/usr/bin/env python
"""Tuning aid / BASELINE config 3: D8 stochastic-MFD accumulation on a 4096^2 DEM
(example/dem_multiflow.py:43-49): random_weighted + accumulate per realisation."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soillib_amd import _abi, silt, soil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--k", type=int, default=8)
args = ap.parse_args()
lib = _abi.lib()
S = args.size
npar = soil.noise_t()
npar.seed = 3.0
npar.ext = [S, S]
h = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
silt.multiply(h, 100.0)
rain = silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
silt.set(rain, 1.0)
flow = soil.random_weighted(h, soil.d8, 0, 0, 10.0)
acc = soil.accumulate(flow, rain, soil.d8)          # warm-up (allocates the workspace)
_abi.check(lib.soil_device_synchronize())
t0 = time.perf_counter()
for k in range(args.k):
    flow = soil.random_weighted(h, soil.d8, 0, k, 10.0)
    acc = soil.accumulate(flow, rain, soil.d8)
_abi.check(lib.soil_device_synchronize())
dt = (time.perf_counter() - t0) / args.k
a = acc.cpu().numpy()
f = flow.cpu().numpy()
print("MFD accumulate %dx%d D8: %.2f ms per realisation = %.0f Mcells/s; outlets sum %.0f (cells %d)" % (
    S, S, dt * 1e3, S * S / dt / 1e6, a[f < 0].sum(), S * S))
