"""A whole world of slabs on ONE GPU, its ranks taking turns (threads, an in-process wire of device copies):
the wall time of a step is the SUM of the ranks' work, the immigrants' launches of the migrate mode
included — what tools/bench_rank_of_world.py leaves out.  For the A/B of the two halo modes' compute:

    python tools/bench_world_on_one_gpu.py --world 4 --grid 16384 --steps 3 [--mode deep|migrate]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--grid", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="deep", choices=("deep", "migrate"))
    args = ap.parse_args()
    from soillib_amd import soil
    from soillib_amd.parallel import CallbackComm, SlabRunner
    from test_gpu_parallel import LocalWire
    from util import script_param
    world, G = args.world, args.grid
    S = G // world
    param = script_param(soil.param_t())
    shared = LocalWire.Shared(world)
    out, errs = [None] * world, []
    t = {}

    def worker(rank):
        held = False
        try:
            shared.token.acquire()
            held = True
            r = SlabRunner(rows_per_rank=S, W=G, param=param, particles_div=8, seed=0, scale=[20.0 / G, 20.0 / G, 4.0],
                           noise_rows=G, comm=CallbackComm(rank, world, LocalWire(shared, rank)), device=0, mode=args.mode)
            for _ in range(args.warmup):
                r.step()
                r.sync()
            r.barrier()
            if rank == 0:
                t["t0"] = time.perf_counter()
            for _ in range(args.steps):
                r.step()
                r.sync()
            r.barrier()
            if rank == 0:
                t["t1"] = time.perf_counter()
            out[rank] = dict(r.migration, rows=r.rows, fallbacks=r.fallbacks)
            r.close()
        except BaseException as e:
            errs.append(e)
            try:
                shared.bar.abort()
            except Exception:
                pass
        finally:
            if held:
                shared.token.release()

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(world)]
    [x.start() for x in ts]
    [x.join() for x in ts]
    if errs:
        raise errs[0]
    ms = (t["t1"] - t["t0"]) / args.steps * 1e3
    n = args.steps + args.warmup
    print("world %d of %d^2, mode %s: %.2f ms per step for ALL ranks = %.2f ms per rank; rows per rank %s; per step: "
          "walkers handed %s, launches %s; repeated launches %s" % (
              world, G, args.mode, ms, ms / world, [o["rows"] for o in out], [o["walkers_handed"] // n for o in out],
              [round(o["passes"] / n, 1) for o in out], [o["fallbacks"] for o in out]), flush=True)


if __name__ == "__main__":
    main()
