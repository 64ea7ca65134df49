"""Worker of tests/test_parallel_gloo.py: one rank of a world_size-N gloo job that drives the
library's slab runner (soil_slab_step, csrc/slab_runner.hip) with the CPU ORACLE plugged in as
compute back-end through soil_slab_ops and gloo as the wire through soil_comm.  The product
back-end is HIP-only; what is under test here is the library's partition, halo-exchange, trimming
and repeat-launch logic — the very code that runs under RCCL on a node."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle import pyoracle as o  # noqa: E402
from soillib_amd import _abi, parallel  # noqa: E402
from util import copy_param, script_param  # noqa: E402


def _arr(addr, shape, dtype=np.float32):
    n = int(np.prod(shape))
    ct = {np.float32: C.c_float, np.int32: C.c_int32}[dtype]
    return np.ctypeslib.as_array((ct * n).from_address(int(addr))).reshape(shape)


class OracleOps:
    """soil_slab_ops on host memory: every entry forwards to the oracle (the checker, used here
    as a stand-in device — tests only)."""

    def __init__(self):
        self.blocks = {}

    # memory
    def alloc(self, out, nbytes):
        buf = np.zeros(max(int(nbytes), 8), np.uint8)
        self.blocks[buf.ctypes.data] = buf
        out[0] = buf.ctypes.data

    def release(self, p):
        self.blocks.pop(int(p or 0), None)

    def fill_f32(self, dst, value, n, lane):
        _arr(dst, (n,))[:] = value

    def add_f32(self, dst, src, n, lane):
        _arr(dst, (n,))[:] += _arr(src, (n,))

    def rng_seed(self, rng, N, seed, offset):
        a = np.ctypeslib.as_array((C.c_uint64 * (2 * N)).from_address(int(rng))).reshape(N, 2)
        a[:, 0] = seed
        a[:, 1] = offset

    @staticmethod
    def _rng(rng, N):
        return np.ctypeslib.as_array((C.c_uint64 * (2 * N)).from_address(int(rng))).view(o.RNG_DTYPE).reshape(N)

    @staticmethod
    def _planes(pl, dom):
        P, d = pl.contents, dom.contents
        sh1, sh2 = (d.rows, d.W), (d.rows, d.W, 2)
        two = ("layers", "layers_next", "velocity", "velocityFlux", "debrisVelocity", "debrisVelocityFlux")
        return {n: _arr(getattr(P, n), sh2 if n in two else sh1) for n in _abi._PLANES}, d

    @staticmethod
    def _param(param):
        return copy_param(param.contents, o.default_param())

    @staticmethod
    def _dom(d, r0=None, r1=None):
        return o.domain(d.H, d.W, d.x0, d.rows, d.r0 if r0 is None else r0, d.r1 if r1 is None else r1)

    def particles_fluvial(self, pl, rng, N, remote0, dom, scale, param):
        P, d = self._planes(pl, dom)
        o.particles_fluvial(P["waterFlux"], P["massFlux"], P["velocityFlux"], None, self._rng(rng, N),
                            P["layers"], P["rainfall"], P["waterHeight"], P["velocity"], None,
                            [scale[i] for i in range(3)], self._param(param), dom=self._dom(d),
                            remote0=_arr(remote0, (8,)))

    def particles_debris(self, pl, rng, N, remote0, dom, scale, param):
        P, d = self._planes(pl, dom)
        o.particles_debris(P["debrisFlux"], P["debrisVelocityFlux"], None, self._rng(rng, N), P["layers"],
                           P["debrisVelocity"], None, [scale[i] for i in range(3)], self._param(param),
                           dom=self._dom(d), remote0=_arr(remote0, (8,)))

    def cells(self, pl, dom, scale, param):
        P, d = self._planes(pl, dom)
        r0, r1 = d.r0, d.r1
        if r1 <= r0:
            return
        res = o.erode_cells(P["layers"], P["uplift"], P["rainfall"], P["waterFlux"], P["massFlux"],
                            P["velocityFlux"], P["debrisFlux"], P["debrisVelocityFlux"],
                            [scale[i] for i in range(3)], self._param(param), dom=self._dom(d))
        for name in ("layers_next", "height", "waterHeight", "mass", "velocity", "debris", "debrisVelocity"):
            P[name][r0:r1] = res[name][r0:r1]
        for name in parallel.FLUX_PLANES:      # the fused kernel re-zeroes what it consumed
            P[name][r0:r1] = 0

    def ghost_extent(self, plane, rows, row_floats, r0, r1, depth):
        a = _arr(plane, (rows, row_floats))
        hit = np.nonzero((a.view(np.uint32) != 0).any(axis=1))[0]
        hit = hit[(hit < r0) | (hit >= r1)]
        if len(hit):
            depth[0] = max(depth[0], int(r0 - hit.min()))
            depth[1] = max(depth[1], int(hit.max()) - r1 + 1)

    def noise_rows(self, out, rows, W, x0, p):
        q = p.contents
        H = int(q.ext[0])
        full = o.noise(H, W, seed=q.seed, ext=(q.ext[0], q.ext[1]))
        _arr(out, (rows, W))[:] = full[x0:x0 + rows]

    def layers_from_bedrock(self, layers, bed, n):
        a = _arr(layers, (n, 2))
        a[:, 0] = _arr(bed, (n,))
        a[:, 1] = 0

    def to_host(self, dst, src, nbytes):
        C.memmove(dst, src, nbytes)

    def from_host(self, dst, src, nbytes):
        C.memmove(dst, src, nbytes)

    def fork(self):
        pass

    def join(self):
        pass

    def sync(self):
        pass

    def stream(self, lane):
        return None


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
    pair = len(sys.argv) > 6 and sys.argv[6] == "pair"
    param = copy_param(script_param(o.default_param()), _abi.Param())
    param.maxage = maxage
    backend = OracleOps()
    if pair:    # both launches "overlapped": the fluvial one draws from rng, the debris one from rng_debris
        def particles_pair(pl, rng, rng_debris, N, remote0, dom, scale, prm):
            backend.particles_fluvial(pl, rng, N, remote0, dom, scale, prm)
            backend.particles_debris(pl, rng_debris, N, remote0, dom, scale, prm)
        backend.particles_pair = particles_pair
    wire = parallel.GlooWire(device=False)
    comm = parallel.CallbackComm(wire.dist.get_rank(), wire.dist.get_world_size(), wire)
    runner = parallel.SlabRunner(rows_per_rank=S, W=W, param=param, particles_div=8, seed=0,
                                 ops=parallel.CallbackOps(backend), comm=comm, pair=pair)
    for _ in range(steps):
        runner.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % runner.rank),
             layers=runner.plane("layers", owned=True).copy(),
             waterHeight=runner.plane("waterHeight", owned=True).copy(),
             velocity=runner.plane("velocity", owned=True).copy(),
             debris=runner.plane("debris", owned=True).copy(),
             height=runner.plane("height", owned=True).copy(),
             ghost_layers=runner.plane("layers").copy(), x0=runner.x0, rows=runner.rows,
             G=runner.G, H=runner.H, fallbacks=runner.fallbacks,
             halo_rows=np.array([runner.halo_rows[k] for k in ("flux", "field", "full", "window", "window_full")]))
    t = runner.max_over_ranks(float(runner.rank))
    assert t == runner.world - 1
    runner.barrier()
    runner.shutdown()


if __name__ == "__main__":
    main()
