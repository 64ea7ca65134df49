"""Worker of tests/test_parallel_gloo.py: one rank of a world_size-N gloo job that
runs soillib_amd.parallel.SlabRunner with the CPU ORACLE as compute back-end
(the product back-end is HIP-only; what is under test here is the partition
and halo-exchange logic, which is back-end independent)."""
import contextlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle import pyoracle as o  # noqa: E402
from soillib_amd import parallel  # noqa: E402
from util import script_param  # noqa: E402


class OracleOps:
    device = "cpu"

    def alloc(self, shape, kind="f32"):
        if kind == "rng":
            return o.rng_seed(shape[0], 0, 0)
        return torch.zeros(tuple(shape), dtype=torch.float32)

    def ghost_rows(self, param):
        return int(math.ceil(1.41421356237309515 * param.maxage)) + 2

    def seed(self, rng, seed, offset):
        rng["seed"] = seed
        rng["offset"] = offset

    def fill(self, t, value):
        t.fill_(value)

    def zero(self, t):
        t.zero_()

    def add(self, dst, src):
        dst.add_(src)

    def noise_rows(self, out, H, W, x0, seed):
        full = o.noise(H, W, seed=seed, ext=(float(H), float(W)))
        out.copy_(torch.from_numpy(full[x0:x0 + out.shape[0]]))

    def layers_from_bedrock(self, layers, bed):
        layers[..., 0] = bed
        layers[..., 1] = 0

    @staticmethod
    def _dom(dom, r0=None, r1=None):
        return o.domain(dom.H, dom.W, dom.x0, dom.rows, dom.r0 if r0 is None else r0,
                        dom.r1 if r1 is None else r1)

    def particles_fluvial(self, P, rng, N, dom, scale, param, remote0):
        o.particles_fluvial(P["waterFlux"].numpy(), P["massFlux"].numpy(),
                            P["velocityFlux"].numpy(), None, rng, P["layers"].numpy(),
                            P["rainfall"].numpy(), P["waterHeight"].numpy(),
                            P["velocity"].numpy(), None, scale, param, dom=self._dom(dom),
                            remote0=remote0.numpy())

    def particles_debris(self, P, rng, N, dom, scale, param, remote0):
        o.particles_debris(P["debrisFlux"].numpy(), P["debrisVelocityFlux"].numpy(), None, rng,
                           P["layers"].numpy(), P["debrisVelocity"].numpy(), None, scale, param,
                           dom=self._dom(dom), remote0=remote0.numpy())

    def ghost_extent(self, planes, r0, r1):
        up = down = 0
        for p in planes:
            a = p.numpy().reshape(p.shape[0], -1)
            rows = np.nonzero((a != 0).any(axis=1))[0]
            if len(rows):
                up = max(up, r0 - int(rows.min()))
                down = max(down, int(rows.max()) - r1 + 1)
        return max(up, 0), max(down, 0)

    def add_cell0(self, P, remote0):
        r = remote0.numpy()
        P["waterFlux"].numpy().ravel()[0] += r[0]
        P["massFlux"].numpy().ravel()[0] += r[1]
        P["velocityFlux"].numpy().ravel()[0:2] += r[2:4]
        P["debrisFlux"].numpy().ravel()[0] += r[4]
        P["debrisVelocityFlux"].numpy().ravel()[0:2] += r[5:7]

    def cells(self, P, dom, r0, r1, scale, param):
        if r1 <= r0:
            return
        d = self._dom(dom, r0, r1)
        res = o.erode_cells(P["layers"].numpy(), P["uplift"].numpy(), P["rainfall"].numpy(),
                            P["waterFlux"].numpy(), P["massFlux"].numpy(),
                            P["velocityFlux"].numpy(), P["debrisFlux"].numpy(),
                            P["debrisVelocityFlux"].numpy(), scale, param, dom=d)
        for name in ("layers_next", "height", "waterHeight", "mass", "velocity", "debris",
                     "debrisVelocity"):
            P[name].numpy()[r0:r1] = res[name][r0:r1]
        for name in parallel.FLUX_PLANES:      # the fused kernel re-zeroes what it consumed
            P[name][r0:r1] = 0

    def fork_comm(self):
        return contextlib.nullcontext()

    def join_comm(self):
        pass

    def sync(self):
        pass


def multiflow_main():
    """parallel.multiflow with the oracle as back-end: realisations sharded over ranks."""
    out_dir, H, W, K = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    torch.distributed.init_process_group("gloo")
    rank = torch.distributed.get_rank()
    dem = o.noise(H, W, seed=1.0, ext=(float(H), float(W))) * 100.0
    rain = np.ones((H, W), np.float32)

    def local_sum(first, stride):
        total = np.zeros((H, W), np.float64)
        for k in range(first, K, stride):
            acc = o.accumulate(o.random_weighted(dem, 1, 0, k, 10.0), rain, 1)
            total += (acc / np.float32(K)).astype(np.float64)
        return torch.from_numpy(total)

    mean = parallel.multiflow(None, None, K, 10.0, local_sum=local_sum)
    np.save(os.path.join(out_dir, "multiflow_rank%d.npy" % rank), mean.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
    if sys.argv[1] == "multiflow":
        return multiflow_main()
    out_dir, S, W, steps, maxage = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), \
        int(sys.argv[4]), int(sys.argv[5])
    param = script_param(o.default_param())
    param.maxage = maxage
    runner = parallel.SlabRunner(rows_per_rank=S, W=W, param=param, particles_div=8, seed=0,
                                 ops=OracleOps())
    for _ in range(steps):
        runner.step()
    own = slice(runner.r0, runner.r1)
    np.savez(os.path.join(out_dir, "rank%d.npz" % runner.rank),
             layers=runner.P["layers"].numpy()[own], waterHeight=runner.P["waterHeight"].numpy()[own],
             velocity=runner.P["velocity"].numpy()[own], debris=runner.P["debris"].numpy()[own],
             height=runner.P["height"].numpy()[own],
             ghost_layers=runner.P["layers"].numpy(), x0=runner.x0, rows=runner.rows,
             G=runner.G, H=runner.H, fallbacks=runner.fallbacks,
             halo_rows=np.array([runner.halo_rows[k] for k in ("flux", "field", "full")]))
    t = runner.max_over_ranks(float(runner.rank))
    assert t == runner.world - 1
    runner.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
