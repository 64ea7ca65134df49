#!/usr/bin/env python
"""Tuning aid: time soil_erode_cells_fused alone (back-to-back launches between two
HIP events) and report GB/s against the 112 B/cell algorithmic traffic."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soillib_amd import _abi, silt, soil  # noqa: E402
from soillib_amd.erosion import ErosionModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--skew", type=int, default=0, help="bytes of extra offset per plane")
ap.add_argument("--arena", action="store_true", help="carve the planes out of big device allocations")
ap.add_argument("--chunk-mb", type=int, default=16384, help="size of one arena allocation (MiB)")
args = ap.parse_args()
lib = _abi.lib()
S = args.size
p = soil.param_t()
_keep = []
_count = [0]


def skew_alloc(dtype, shape):
    n = 1
    for d in shape:
        n *= d
    nbytes = n * dtype.itemsize
    off = args.skew * _count[0]
    _count[0] += 1
    raw = silt.tensor(silt.float32, silt.shape((nbytes + off + 4095) // 4 + 1024), silt.gpu)
    _keep.append(raw)
    return silt.tensor.from_device(raw.ptr + off, dtype, silt.shape(*shape), keepalive=raw)


_arena = {"raw": None, "off": 0, "all": []}


def arena_alloc(dtype, shape):
    n = 1
    for d in shape:
        n *= d
    chunk = args.chunk_mb * 1024 * 1024
    nbytes = (n * dtype.itemsize + 2097151) // 2097152 * 2097152 + args.skew
    if _arena["raw"] is None or _arena["off"] + nbytes > chunk:
        _arena["raw"] = silt.tensor(silt.float32, silt.shape(chunk // 4), silt.gpu)
        _arena["all"].append(_arena["raw"])
        _arena["off"] = 0
    off = _arena["off"]
    _arena["off"] += nbytes
    return silt.tensor.from_device(_arena["raw"].ptr + off, dtype, silt.shape(*shape),
                                   keepalive=_arena["raw"])


m = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), p, 1024,
                 alloc=arena_alloc if args.arena else (skew_alloc if args.skew else None))
npar = soil.noise_t()
npar.seed = 3.0
npar.ext = [S, S]
bed = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
_abi.check(lib.soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, None, bed.elem(), None))
silt.set(m.rainfall, 1.0)
ev = [C.c_void_p(), C.c_void_p()]
for e in ev:
    _abi.check(lib.soil_event_create(C.byref(e)))
for rep in range(3):
    for _ in range(5):
        m.cells_fused(); m.swap_layers()
    _abi.check(lib.soil_event_record(ev[0], None))
    for _ in range(args.iters):
        m.cells_fused(); m.swap_layers()
    _abi.check(lib.soil_event_record(ev[1], None))
    ms = C.c_float()
    _abi.check(lib.soil_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
    t = ms.value / args.iters
    print("fused cells %dx%d: %.4f ms/launch  %.1f GB/s (112 B/cell)  %.1f%% of 8 TB/s" % (
        S, S, t, 112 * S * S / t / 1e6, 112 * S * S / t / 1e6 / 80.0))
