#!/usr/bin/env python
"""BASELINE config 3: D8 stochastic multiple-flow accumulation on a 4096^2 DEM
(example/dem_multiflow.py:43-49): K realisations of random_weighted + accumulate,
averaged on the device (soil_multiflow).  `--gpus N` under torch.distributed.run
shards the realisations over ranks (soillib_amd.parallel.multiflow)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soillib_amd import _abi, parallel, silt, soil  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--k", type=int, default=32)
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    torch.distributed.init_process_group("nccl")
lib = _abi.lib()
S = args.size
npar = soil.noise_t()
npar.seed = 3.0
npar.ext = [S, S]
h = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
silt.multiply(h, 100.0)
rain = silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
silt.set(rain, 1.0)
if rank == 0:                                        # pit fill of the same DEM (config 3's other half)
    soil.fill_depressions(h, soil.d8)
    _abi.check(lib.soil_device_synchronize())
    t0 = time.perf_counter()
    filled = soil.fill_depressions(h, soil.d8)
    _abi.check(lib.soil_device_synchronize())
    dt = time.perf_counter() - t0
    a, b = filled.cpu().numpy(), h.cpu().numpy()
    print("fill_depressions %dx%d D8: %.2f ms; %d cells raised, max %.3f" % (
        S, S, dt * 1e3, int((a > b).sum()), float((a - b).max())))
    h = filled
parallel.multiflow(h, rain, world, 10.0)            # warm-up (allocates the workspaces)
_abi.check(lib.soil_device_synchronize())
t0 = time.perf_counter()
mean = parallel.multiflow(h, rain, args.k, 10.0)
_abi.check(lib.soil_device_synchronize())
dt = time.perf_counter() - t0
if rank == 0:
    m = mean.cpu().numpy()
    print("MFD accumulate %dx%d D8, K=%d on %d GPU(s): %.2f ms per realisation = %.0f Mcells/s "
          "(whole job); mean upstream area min %.3f max %.0f" % (
              S, S, args.k, world, dt / args.k * 1e3, S * S * args.k / dt / 1e6, m.min(), m.max()))
if world > 1:
    torch.distributed.destroy_process_group()
