#!/usr/bin/env python
"""Achieved HBM bandwidth of the per-cell kernels next to their algorithmic bytes
(steepest/direction 8, random_weighted 8, gradient 12, negslope 8,
laplacian 8*D, gaussian_blur 2 passes x 8*C, normal 16 B/cell) on an S^2 grid."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soillib_amd import _abi, silt, soil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
S = args.size
lib = _abi.lib()
p = soil.noise_t()
p.seed = 3.0
p.ext = [S, S]
h = soil.noise(silt.shape(S, S), p, host=silt.gpu)
silt.multiply(h, 100.0)
v2 = silt.tensor(silt.float32, silt.shape(S, S, 2), silt.gpu)
silt.set(v2, 1.0)
flow = soil.steepest(h, soil.d8)


def ev():
    e = __import__("ctypes").c_void_p()
    _abi.check(lib.soil_event_create(__import__("ctypes").byref(e)))
    return e


def timed(fn):
    fn()
    a, b = ev(), ev()
    _abi.check(lib.soil_event_record(a, None))
    for _ in range(args.reps):
        fn()
    _abi.check(lib.soil_event_record(b, None))
    ms = __import__("ctypes").c_float()
    _abi.check(lib.soil_event_elapsed_ms(a, b, __import__("ctypes").byref(ms)))
    return ms.value / args.reps


cells = S * S
cases = [
    ("steepest d8", 8, lambda: soil.steepest(h, soil.d8)),
    ("direction d8", 8, lambda: soil.direction(h, soil.d8)),
    ("random_weighted d8", 8, lambda: soil.random_weighted(h, soil.d8, 0, 1, 10.0)),
    ("slope", 12, lambda: soil.slope(h, flow, (1.0, 1.0))),
    ("gradient", 12, lambda: soil.gradient(h, (1.0, 1.0))),
    ("negslope", 8, lambda: soil.negslope(h, (1.0, 1.0))),
    ("laplacian D=1", 8, lambda: soil.laplacian(h, (1.0, 1.0))),
    ("laplacian D=2", 16, lambda: soil.laplacian(v2, (1.0, 1.0))),
    ("gaussian_blur C=1", 16, lambda: soil.gaussian_blur(h, 3.0)),
    ("gaussian_blur C=2", 32, lambda: soil.gaussian_blur(v2, 3.0)),
    ("normal", 16, lambda: soil.normal(h, (1.0, 1.0, 1.0))),
    ("fill_depressions d8 (iterative)", 8, lambda: soil.fill_depressions(h, soil.d8)),
]
print("%dx%d, %d reps (allocation of the result included, as the API returns new tensors)" % (S, S, args.reps))
for name, bytes_per_cell, fn in cases:
    ms = timed(fn)
    gbs = bytes_per_cell * cells / (ms * 1e-3) / 1e9
    print("%-32s %8.3f ms  %7.0f GB/s algorithmic = %4.1f %% of 8 TB/s  (%d B/cell)" % (
        name, ms, gbs, gbs / 80.0, bytes_per_cell))
