"""Worker of tests/test_gpu_parallel.py::test_two_processes_share_one_gpu_over_gloo: one rank of
a SlabRunner world whose ranks are separate processes on the SAME GPU, talking through
torch.distributed's gloo backend (SOIL_DEVICE=0, SOIL_DIST_BACKEND=gloo)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, S, W, maxage, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    sample = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else None   # owned rows to keep
    from soillib_amd import parallel, soil
    from util import script_param
    p = script_param(soil.param_t())
    p.maxage = maxage
    r = parallel.SlabRunner(rows_per_rank=S, W=W, param=p, particles_div=8, seed=0)
    for _ in range(steps):
        r.step()
    r.sync()
    planes = {k: r.plane(k, owned=True) for k in ("layers", "waterHeight", "velocity", "debris")}
    if sample is not None:   # (a 2048 x 16384 slab: a few rows per rank travel back, not 800 MB)
        planes = {k: np.ascontiguousarray(v[sample]) for k, v in planes.items()}
        planes["halo_rows"] = np.array([r.halo_rows["flux"], r.halo_rows["field"], r.halo_rows["full"]], np.int64)
        planes["fallbacks"] = np.array([r.fallbacks], np.int64)
    np.savez(os.path.join(out_dir, "rank%d.npz" % r.rank), **planes)
    assert r.max_over_ranks(float(r.rank)) == r.world - 1
    r.shutdown()


if __name__ == "__main__":
    main()
