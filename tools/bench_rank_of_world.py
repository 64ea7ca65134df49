"""What ONE rank of a `world`-GPU run computes per step, measured on a single GPU.

The sharded step (soillib_amd.parallel.SlabRunner) is run as an interior rank of an
emulated world with a communicator that moves nothing: every kernel the rank would
launch runs at its real size (slab + ghost rows, all world*N particle streams
replayed, halo adds), only the wire time is missing.  Dividing the world = 1 time by
these gives the compute-side ceiling of the weak-scaling efficiency bench.py --gpus N
can reach; what xGMI adds comes on top.

    python tools/bench_rank_of_world.py [--worlds 1,2,4,8] [--size 8192] [--steps 4]
"""
import argparse
import os
import sys
import time

# the stand-in communicator moves nothing: the ghost fields a rank walks on are never refreshed (the
# timing does not care).  SOIL_HALO_FULL=0 in the environment turns the trimming by measured reach on
# (this rank's own reach; the launch window of round 4 with it, SOIL_HALO_WINDOW=0 without)
os.environ.setdefault("SOIL_HALO_FULL", "1")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


class NullWire:
    """impl of parallel.CallbackComm that moves nothing"""

    def exchange(self, sends, recvs):
        pass

    def all_reduce(self, addr, n):
        pass

    def barrier(self):
        pass

    def max_over_ranks(self, value):
        return value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--strong", type=int, default=0,
                    help="strong scaling: ONE grid of this size cut into `world` slabs (BASELINE config 5: "
                         "16384); the figure to compare with is the same grid on one GPU "
                         "(python bench.py --size 16384)")
    ap.add_argument("--mode", default="deep", choices=("deep", "migrate"),
                    help="deep halos, or walkers handed over at a shallow halo's end (soil_slab.h).  With this "
                         "wire nobody hands this rank anything: the time lacks the launches that walk the "
                         "immigrants on (about as many walkers as `handed` says left)")
    args = ap.parse_args()
    from soillib_amd import soil
    from soillib_amd.parallel import CallbackComm, SlabRunner
    from util import script_param
    base = None
    for world in [int(w) for w in args.worlds.split(",")]:
        param = script_param(soil.param_t())
        if args.strong:
            S, Wc, cell = args.strong // world, args.strong, 20.0 / args.strong
        else:
            S, Wc, cell = args.size, args.size, 20.0 / args.size
        r = SlabRunner(rows_per_rank=S, W=Wc, param=param, particles_div=8, seed=0,
                       comm=CallbackComm(world // 2, world, NullWire()),
                       scale=[cell, cell, 4.0], noise_rows=Wc, mode=args.mode)
        for _ in range(args.warmup):
            r.step()
        r.sync()
        soil.particle_steps(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r.step()
        r.sync()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        steps = soil.particle_steps(reset=True) / args.steps
        base = base or ms
        print("world %d rank %d: rows %d (+%d ghost), N %d: %.2f ms/step, %.2f G particle steps, "
              "compute-side efficiency %.3f; repeated launches %d, ghost rows walked / bound %.2f; mode %s %s" % (
                  world, r.rank, S, r.rows - S, r.N, ms, steps / 1e9,
                  base / ms / (world if args.strong else 1), r.fallbacks,
                  r.halo_rows["window"] / max(r.halo_rows["window_full"], 1), r.mode,
                  {k: v // (args.steps + args.warmup) for k, v in r.migration.items()}), flush=True)
        r.close()
        del r
        from soillib_amd import silt
        silt.empty_cache()


if __name__ == "__main__":
    main()
