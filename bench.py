#!/usr/bin/env python
"""bench.py — Mcells/s of the hydraulic-erosion step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full erosion step of the hot path over the resident grid
(SURVEY.md §3.1): re-seed the particle streams, fluvial + debris Monte-Carlo
transport (N = H*W/8 particles, maxage 256), then the fused per-cell phase
(normalise x2, mass transfer, creep, layer update, merge, flux re-zero).
Workload at N=1: BASELINE.json configs[3], the 8192^2 coupled hydraulic +
thermal step on synthetic OpenSimplex2-FBm terrain (soil.noise, generated on
the device), inputs resident in HBM before the timed region.  `--gpus N` > 1 starts
its N ranks itself (torch.distributed.run on 127.0.0.1, one per GPU) and runs the
library's slab runner (soil_slab_step, include/soil_slab.h: row slabs, deep halos
trimmed to the measured reach, RCCL send/recv groups on the library's streams):
`value` is weak scaling, one 8192-row slab per GPU (global grid (N*8192) x 8192),
and the line carries `strong16384`, BASELINE.json configs[4] as written — the
16384^2 grid cut into N row slabs — with `speedup_vs_1gpu` measured in the same run.
`--grid 16384` (or SOIL_BENCH_GRID=16384) makes that strong-scaling point the
line's own `value` ("scaling": "strong"; SOIL_BENCH_FORCE_SLAB=1 runs the N=1
point through the same slab code).

Prints ONE JSON line (rank 0).  `value` is whole-job throughput of the FULL
step; the per-phase split, the roofline of the HBM-bound fused cell kernel
(112 algorithmic bytes per cell) and the CPU baseline (the
oracle, oracle/liboracle.so, on the host cores) ride along in the same line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CELL_BYTES = 112               # fused floor, SURVEY.md §8d / DESIGN.md §Roofline
CELL_BYTES_LAZY = 84           # ... without the 28 B/cell re-zero of the flux planes (lazy chain of steps)


def script_param(soil):
    """example/erosion_gpu.py:75-100 mapped onto the live param_t names."""
    p = soil.param_t()
    p.timeStep = 1000.0
    p.maxage = 256
    p.lrate = 1.0
    p.gravity = 9.81
    p.uplift = 0.01
    p.rainfall = 1.0
    p.evapRate = 0.0005
    p.viscosityWater = 0.000001
    p.bedShearWater = 12.5
    p.suspensionRateFluvial = 0.0008
    p.depositionRateFluvial = 0.00001
    p.fluvialExponent = 0.01
    p.exitSlope = 0.025
    p.critSlopeBedrock = 0.57
    p.landslideRateDebris = 0.0025
    p.suspensionRateDebris = 0.00025
    p.depositionRateDebris = 0.0001
    p.yieldStress = 2E6
    p.densityDebris = 2500.0
    p.viscosityDebris = 0.004
    p.bedShearDebris = 60 / 2500.0
    return p


class Events:
    """HIP events on the launch stream (the C ABI's null stream by default)."""

    def __init__(self, abi, n):
        self.abi, self.lib = abi, abi.lib()
        self.ev = []
        for _ in range(n):
            e = C.c_void_p()
            abi.check(self.lib.soil_event_create(C.byref(e)))
            self.ev.append(e)

    def record(self, i, stream=None):
        self.abi.check(self.lib.soil_event_record(self.ev[i], stream if stream is not None else self.abi.stream()))

    def ms(self, i, j):
        out = C.c_float()
        self.abi.check(self.lib.soil_event_elapsed_ms(self.ev[i], self.ev[j], C.byref(out)))
        return out.value


def usable_cores():
    """Hardware threads this process may really use: the affinity mask, cut down to the cgroup's
    CPU quota when there is one (a container on a 256-thread host may own a handful)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(size, seconds_hint=20.0):
    """Times the oracle (oracle/liboracle.so, the CPU restatement of the same
    kernels) on the host cores on a bounded sample of the same workload."""
    import numpy as np
    from oracle import pyoracle as o
    H = W = size
    N = H * W // 8
    p = o.default_param()
    sp = script_param(__import__("soillib_amd.soil", fromlist=["soil"]))
    for name, _ in p._fields_:
        if name not in ("force", "_pad"):
            setattr(p, name, getattr(sp._c, name))
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0] = o.noise(H, W, seed=3.0, ext=(float(H), float(W)))
    rain = np.ones((H, W), np.float32)
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)

    def one_step(state, step, threads):
        rng = o.rng_seed(N, 0, step * N)
        wf, mf, vf, df, dvf = z1(), z1(), z2(), z1(), z2()
        s1 = o.particles_fluvial(wf, mf, vf, None, rng, state["layers"], rain, state["wh"],
                                 state["v"], None, scale, p, threads=threads)
        s2 = o.particles_debris(df, dvf, None, rng, state["layers"], state["dv"], None, scale, p,
                                threads=threads)
        r = o.erode_cells(state["layers"], z1(), rain, wf, mf, vf, df, dvf, scale, p)
        return dict(layers=r["layers_next"], wh=r["waterHeight"], v=r["velocity"],
                    dv=r["debrisVelocity"]), s1 + s2

    state = dict(layers=layers, wh=z1(), v=z2(), dv=z2())
    o.set_threads(1)
    t0 = time.perf_counter()
    state, steps1 = one_step(state, 0, 1)
    t_single = time.perf_counter() - t0
    # How many threads pay: the box reports its hardware threads, a container may be allowed a
    # fraction of them, and the particle loops stop scaling where their atomic deposits collide.
    # One step per candidate count (thread teams spun up by an untimed step first), best one kept.
    avail = usable_cores()
    cand = sorted({c for c in (2, 4, 8, 16, 32, 64, 128, 256, avail) if 2 <= c <= avail})
    probe = {}
    for c in cand:
        o.set_threads(c)                      # per-cell loops: OpenMP over rows
        one_step(dict(state), 1, c)
        t0 = time.perf_counter()
        one_step(dict(state), 1, c)
        probe[c] = time.perf_counter() - t0
        if probe[c] > 2.0 * min(probe.values()):
            break                             # well past the knee
    cores = min(probe, key=probe.get) if probe else 1
    if probe and probe[cores] > t_single:
        cores = 1
    o.set_threads(cores)
    t_probe = probe.get(cores, t_single)
    reps = max(2, min(16, int(seconds_hint / max(t_probe, 0.05))))
    t0 = time.perf_counter()
    psteps = 0
    for i in range(reps):
        state, s = one_step(state, 1 + i, cores)
        psteps += s
    t_multi = (time.perf_counter() - t0) / reps
    o.set_threads(1)
    return {
        "value": H * W / t_multi / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port",
        "sample": "%dx%d grid, N=%d particles, maxage 256, full step; 1 step on 1 thread "
                  "(%.2f Mcells/s) + %d steps with OpenMP on %d threads (particle loops over "
                  "particles with atomic deposits, per-cell loops over rows); thread count chosen "
                  "by a one-step probe of %s of the %d usable hardware threads" % (
                      H, W, N, H * W / t_single / 1e6, reps, cores,
                      "/".join(str(c) for c in sorted(probe)), avail),
        "threads_probe_s": {str(c): round(t, 4) for c, t in sorted(probe.items())},
        "value_1thread": H * W / t_single / 1e6,
        "mparticle_steps_per_s": psteps / reps / t_multi / 1e6,
    }


def _respawn(n):
    """`python bench.py --gpus N` from a bare shell: start the N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1 and hand their output through; rank 0 prints the line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


ARITH_NOTE = {
    "exact": "IEEE quotients and square root in the particle step (soil_set_particle_arith(0)): the reference "
             "build's arithmetic, trajectories equal to the oracle's step for step (the mode of the -m gpu parity tests)",
    "fast": "quotients of the particle step as numerator x v_rcp_f32(denominator), v_sqrt_f32 "
            "(soil_set_particle_arith(1)): LESS accurate than the reference's own arithmetic; statistical parity "
            "with the oracle only — plane sums 2e-3, visited cells 0.5 %, step counts 0.5 % (tests/test_fast_particles.py)",
}

DEBRIS_NOTE = {
    1: "spent debris walkers end their walks (soil_set_debris_retire(1), the default): with these parameters a debris "
       "walker's attenuations underflow to exact zeros in its first step and the reference walks it through the "
       "other ~100 adding +-0; the product ends a walk once that is certain (include/soil_hip.h) — same flux planes "
       "(x + 0 = x), tests/test_debris_retire.py; the whole GPU suite runs with the walkers marked and watched and "
       "counts no deposit from a marked one.  `debris_walked_to_the_end` times the same steps without it",
    0: "every debris walker walked to the end (soil_set_debris_retire(0)), as the reference's kernel does",
    2: "spent debris walkers marked, walked on and watched (soil_set_debris_retire(2): the test suite's mode)",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=("c1", "c2", "c3", "c4"), default="c4",
                    help="which of BASELINE.json's configs the line is for: c4 (default) = configs[3], the 8192^2 "
                         "coupled step the metric is quoted on (and configs[4] with --gpus N / --grid); c2 = configs[1], "
                         "1024^2 x 10 000 steps (example/erosion_gpu.py:75-106), this same step loop; c1 = configs[0], "
                         "256^2 GeoTIFF -> host normal map; c3 = configs[2], 4096^2 D8 multi-flow accumulation, "
                         "K = --steps realisations (512) — bench_configs.py.  One JSON line each")
    ap.add_argument("--steps", type=int, default=None, help="default: 10 (c4), 10000 (c2), 200 (c1), 512 = K (c3)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 2 (c4), 10 (c2), 5 (c1), 8 (c3)")
    ap.add_argument("--size", type=int, default=None, help="rows per GPU and columns (8192; c2: 1024; c3: 4096)")
    ap.add_argument("--grid", type=int, default=int(os.environ.get("SOIL_BENCH_GRID", "0")),
                    help="STRONG scaling on a fixed grid x grid domain as the line's `value` "
                         "(BASELINE.json configs[4]: 16384): every rank takes grid/N rows of all "
                         "grid columns.  Default 0: `value` is weak scaling, one --size-row slab of "
                         "--size columns per GPU, and a run on more than one GPU appends the "
                         "strong-scaling point of --strong-grid as `strong16384`")
    ap.add_argument("--strong-grid", type=int,
                    default=int(os.environ.get("SOIL_BENCH_STRONG_GRID", "16384")),
                    help="grid of the strong-scaling block a multi-GPU run appends (0: none)")
    ap.add_argument("--particles-div", type=int, default=8, help="N = cells / this")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=1024)
    ap.add_argument("--overlap-particles", action="store_true",
                    help="(the default everywhere since round 3; kept for old command lines)")
    ap.add_argument("--sequential-particles", action="store_true",
                    help="the two particle launches back to back, each timed by itself")
    ap.add_argument("--particle-arith", choices=("fast", "exact"),
                    default=os.environ.get("SOIL_BENCH_ARITH", "exact"),
                    help="arithmetic of the particle step (soil_set_particle_arith): 'exact' (default, the mode of "
                         "the line's `value`) = IEEE quotients and square root as the reference's CUDA build has "
                         "them (no -use_fast_math, CMakeLists.txt:19), the oracle's walks step for step — the mode "
                         "of the parity tests; 'fast' = v_rcp_f32 quotients, statistical parity only "
                         "(tests/test_fast_particles.py).  A single-GPU run also times the OTHER mode for a few "
                         "steps and reports it as the side block `fast_arithmetic` / `exact_arithmetic`")
    ap.add_argument("--halo-mode", choices=("deep", "migrate"), default=os.environ.get("SOIL_BENCH_HALO_MODE", "deep"),
                    help="multi-GPU runs: how a walk that crosses a slab's edge is served (include/soil_slab.h): "
                         "'deep' halos of ceil(sqrt(2) maxage) + 2 rows trimmed to the measured reach, or 'migrate': a "
                         "64-row halo, walkers handed over at its far end as 64-byte records")
    ap.add_argument("--particle-mode", type=int, default=0,
                    help="0 auto, 1 direct (reference launch shape), 2 staged — ablation")
    args = ap.parse_args()
    args.steps_given, args.warmup_given, args.size_given = args.steps is not None, args.warmup is not None, args.size is not None
    dflt = {"c4": (10, 2, 8192), "c2": (10000, 10, 1024), "c1": (200, 5, 256), "c3": (512, 8, 4096)}[args.config]
    if args.steps is None:
        args.steps = dflt[0]
    if args.warmup is None:
        args.warmup = dflt[1]
    if args.size is None:
        args.size = dflt[2]
    return args


class _Single:
    """The whole grid on this GPU: the phases of soil_erode_step issued one by one so that HIP
    events can be put between them."""

    def __init__(self, H, W, param, particles_div, serial):
        from soillib_amd import _abi, silt, soil
        from soillib_amd.erosion import ErosionModel
        self.abi, self.lib, self.serial = _abi, _abi.lib(), serial
        # a chain of steps does not re-zero the flux planes between steps: the particle launches'
        # first rounds overwrite them (soil_erode_step_ex's lazy flags, what soil_erode runs with);
        # SOIL_BENCH_EAGER_FLUX=1: the reference's set(track.*, 0) semantics after every step
        self.lazy = not serial and os.environ.get("SOIL_BENCH_EAGER_FLUX") != "1"
        self.dirty = False
        self.H, self.W = H, W
        scale = (20.0 / H, 20.0 / W, 4.0)
        self.model = model = ErosionModel(H, W, scale, param, H * W // particles_div, seed=0)
        npar = soil.noise_t()
        npar.seed = 3.0
        npar.ext = [H, W]
        bed = soil.noise(silt.shape(H, W), npar, host=silt.gpu)
        # layers[..., 0] = bedrock noise, layers[..., 1] = 0 sediment
        _interleave(self.lib, _abi, model.layers, bed)
        silt.set(model.rainfall, 1.0)
        silt.set(model.uplift, 0.0)

    def step(self, ev=None):
        model = self.model
        model.seed_step()
        if ev: ev.record(0)
        if self.serial:
            model.particles_fluvial()
            if ev: ev.record(1)
            model.particles_debris()
        else:                     # both launches overlapped on two streams
            model.particles_pair(overwrite=self.dirty)
            if ev: ev.record(1)
        if ev: ev.record(2)
        model.cells_fused(keep_flux=self.lazy)
        self.dirty = self.lazy
        if ev: ev.record(3)
        model.swap_layers()
        model.step_index += 1

    def sync(self):
        self.abi.check(self.lib.soil_device_synchronize())

    def barrier(self):
        pass


def timed_steps(runner, ev, steps, warmup, world):
    """W untimed steps, then exactly K steps between barrier + device synchronise on both sides;
    the elapsed wall time is the maximum over the ranks."""
    from soillib_amd import soil
    for _ in range(warmup):
        runner.step()
    runner.barrier()
    runner.sync()
    soil.particle_steps(reset=True)
    phase = [0.0, 0.0, 0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step(ev)
        # events are read back after the step's work is queued; elapsed_ms
        # synchronises on the last event, so this also paces the host
        phase[0] += ev.ms(0, 1)
        phase[1] += ev.ms(1, 2)
        phase[2] += ev.ms(2, 3)
        if world > 1:   # what of the two halo exchanges is not hidden behind a kernel
            phase[3] += ev.ms(2, 4)
            phase[4] += ev.ms(5, 3)
    runner.sync()
    runner.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = runner.max_over_ranks(elapsed)
    return elapsed, phase, soil.particle_steps(reset=True)


def slab_runner(S, Wcols, strong, param, particles_div, comm, pair, mode="deep"):
    """One rank of the library's slab runner (soil_slab_create: csrc/slab_runner.hip)."""
    from soillib_amd import parallel
    # weak scaling: every slab is a piece of the same kind of landscape — cell size
    # (20/S) and noise wavelength per cell as at N = 1, the domain just gets longer.
    # strong scaling: the same grid x grid landscape whatever the world size.
    return parallel.SlabRunner(rows_per_rank=S, W=Wcols, param=param,
                               particles_div=particles_div, seed=0,
                               scale=[20.0 / Wcols, 20.0 / Wcols, 4.0],
                               noise_rows=Wcols if strong else S, comm=comm, pair=pair, mode=mode)


def halo_report(runner):
    hr = runner.halo_rows
    return {"mode": runner.mode, "migration": runner.migration, "ghost_rows_bound": runner.G,
            "rows_shipped_vs_bound": (hr["flux"] + hr["field"]) / max(hr["full"], 1),
            "ghost_rows_walked_vs_bound": hr["window"] / max(hr["window_full"], 1),
            "reach_rows_last_steps": runner.reach_hist, "repeated_launches": runner.fallbacks,
            "trimmed_by_measured_reach": bool(runner.trim)}


def main():
    """A wire that fails or times out (SOIL_ERR_COMM: the library aborts a transfer that does not complete
    within SOIL_RCCL_TIMEOUT_S, include/soil_slab.h) ends the run with ONE JSON line that says so and a
    non-zero exit code — not a process that sits until the driver kills it."""
    from soillib_amd import _abi
    try:
        _main()
    except _abi.CommError as e:
        rank = int(os.environ.get("RANK", "0"))
        print("[bench rank %d] the wire failed: %s" % (rank, e), file=sys.stderr, flush=True)
        if rank == 0:
            args = parse_args()
            print(json.dumps({"metric": "Mcells/s on %d^2 hydraulic-erosion step" % (args.grid or args.size), "value": None,
                              "unit": "Mcells/s", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                              "nccl_ranks": {"backend": "TIMEOUT", "error": str(e)}}), flush=True)
        os._exit(3)   # (an aborted communicator: no orderly teardown of the libraries)


def _main():
    args = parse_args()
    if args.config in ("c1", "c3"):
        if args.gpus > 1:
            raise SystemExit("--config %s is a single-GPU line (c3's realisations shard over ranks: "
                             "tools/bench_accumulate.py --gpus N)" % args.config)
        import bench_configs
        from soillib_amd import _abi
        if args.config == "c3" or _abi.lib().soil_device_count() > 0:
            _abi.check(_abi.lib().soil_set_device(int(os.environ.get("SOIL_DEVICE", "0"))))
        line = (bench_configs.run_c1 if args.config == "c1" else bench_configs.run_c3)(args)
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand or by a driver as plain `python bench.py --gpus N`
        raise SystemExit(_respawn(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SOIL_DEVICE: several ranks on one GPU (functional tests of the multi-rank path over gloo)
    local_rank = int(os.environ.get("SOIL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    args.gpus = world

    from soillib_amd import _abi, silt, soil
    lib = _abi.lib()
    _abi.check(lib.soil_set_device(local_rank))
    _abi.check(lib.soil_set_particle_mode(args.particle_mode))
    soil.particle_arith(args.particle_arith)

    S = args.size
    strong = args.grid > 0
    if strong:
        if args.grid % world:
            raise SystemExit("--grid %d does not split into %d equal slabs" % (args.grid, world))
        S = args.grid // world                   # rows per rank; the columns stay args.grid
    Wcols = args.grid if strong else S
    if args.overlap_particles:
        os.environ["SOIL_STEP_PAIR"] = "1"
    param = script_param(soil)
    slabbed = world > 1 or os.environ.get("SOIL_BENCH_FORCE_SLAB") == "1"
    # one GPU: the two particle launches overlapped, as the library's step driver runs them
    # (soil_erode_step); --sequential-particles for the per-launch phase timings
    # (the slab runner of a multi-GPU run does the same since round 3: a rank of an 8-GPU world
    # 38.0 -> 35.7 ms per step in the single-GPU emulation, tools/bench_rank_of_world.py)
    serial = args.sequential_particles or os.environ.get("SOIL_STEP_PAIR") == "0"
    comm = None
    if slabbed:
        from soillib_amd import parallel
        comm = parallel.default_comm(device=True)      # RCCL inside the library; gloo by SOIL_DIST_BACKEND
        runner = slab_runner(S, Wcols, strong, param, args.particles_div, comm, not serial, args.halo_mode)
        H_global, W = runner.H, Wcols
    else:
        H_global, W = S, Wcols
        runner = _Single(H_global, W, param, args.particles_div, serial)

    ev = Events(_abi, 6)
    elapsed, phase, psteps_rank = timed_steps(runner, ev, args.steps, args.warmup, world)
    other_block, other = None, ("fast" if args.particle_arith == "exact" else "exact")
    if world == 1 and not slabbed and os.environ.get("SOIL_BENCH_NO_OTHER_ARITH", os.environ.get("SOIL_BENCH_NO_EXACT")) != "1":
        # the same model a few steps on in the other arithmetic, same harness — a side block, never `value`
        soil.particle_arith(other)
        ke = max(2, min(args.steps, 6))
        e_el, e_ph, e_ps = timed_steps(runner, ev, ke, 1, world)
        soil.particle_arith(args.particle_arith)
        other_block = {"ms_per_step": e_el / ke * 1e3, "value": H_global * W / (e_el / ke) / 1e6, "unit": "Mcells/s",
                       "steps": ke, "warmup": 1, "gparticle_steps_per_s": e_ps / e_el / 1e9,
                       "note": ARITH_NOTE[other]}
    walked_block = None
    retire_mode = soil.debris_retire()
    if (world == 1 and not slabbed and retire_mode == 1 and
            os.environ.get("SOIL_BENCH_NO_OTHER_ARITH", os.environ.get("SOIL_BENCH_NO_EXACT")) != "1"):
        # the same model a few steps on with every debris walker walked to the end, as the reference walks them
        # (soil_set_debris_retire(0)): the same flux planes, the work the default leaves out — a side block
        soil.debris_retire(0)
        kw = max(2, min(args.steps, 6))
        w_el, w_ph, w_ps = timed_steps(runner, ev, kw, 1, world)
        soil.debris_retire(1)
        walked_block = {"ms_per_step": w_el / kw * 1e3, "value": H_global * W / (w_el / kw) / 1e6, "unit": "Mcells/s",
                        "steps": kw, "warmup": 1, "particle_steps_per_step": w_ps // kw,
                        "gparticle_steps_per_s": w_ps / w_el / 1e9,
                        "note": "soil_set_debris_retire(0): every debris walker walked through all of its steps, as the "
                                "reference's kernel does (erosion.cu:306-349) — the same flux planes as the line's own "
                                "mode (tests/test_debris_retire.py), 0.85 G more particle steps of exact zeros per step"}
    final = None
    if not slabbed:
        # sanity of the evolved terrain (outside the timed region): no NaN/inf may appear
        import numpy as np
        hh = runner.model.height.cpu().numpy()
        final = {"height_min": float(np.nanmin(hh)), "height_max": float(np.nanmax(hh)),
                 "nonfinite_cells": int((~np.isfinite(hh)).sum())}

    probe = None
    if rank == 0:
        # what a plain stream reaches on THIS box (outside the timed region): t += o over two
        # 8192^2 planes, 12 bytes per element.  Boxes of the pool differ by ~25 % here,
        # and the fused cell kernel moves with them.
        a = silt.tensor(silt.float32, silt.shape(8192, 8192), silt.gpu)
        b = silt.tensor(silt.float32, silt.shape(8192, 8192), silt.gpu)
        silt.set(a, 0.0)
        silt.set(b, 1.0)
        pe = Events(_abi, 2)
        for i in range(13):
            if i == 3:
                pe.record(0)
            silt.add(a, b)
        pe.record(1)
        _abi.check(lib.soil_device_synchronize())
        probe = {"kernel": "k_add (t += o), 2 x 8192^2 f32 planes, 12 B/element",
                 "achieved": 12.0 * 8192 * 8192 * 10 / (pe.ms(0, 1) * 1e-3) / 1e9, "unit": "GB/s"}
        del a, b
    halo = halo_report(runner) if world > 1 else None
    nccl_ranks = comm.describe() if slabbed else None    # what RCCL itself reports (ncclCommCount ...)

    # ---- BASELINE.json configs[4]: the strong-scaling point, appended to a multi-GPU weak run ----
    strong_block = None
    G = args.strong_grid
    if world > 1 and not strong and G > 0:
        strong_block = strong_scaling_block(args, runner, ev, rank, world, param, G, comm, not serial)
    if slabbed:
        runner.shutdown()
    if rank != 0:
        return
    K = args.steps
    cells = H_global * W
    ms_step = elapsed / K * 1e3
    cells_rank = S * W
    t_cells = (phase[2] - phase[3] - phase[4]) / K * 1e-3
    # bytes the fused kernel has to move as launched: 112 per cell with the flux planes re-zeroed in
    # the same pass (SURVEY.md 8d's fused floor), 84 when the chain of steps leaves that to the
    # particle launches' first rounds (soil_erode_step_ex: 28 B/cell of zero stores never written)
    lazy = bool(getattr(runner, "lazy", False))
    cell_bytes = CELL_BYTES_LAZY if lazy else CELL_BYTES
    achieved = cell_bytes * cells_rank / t_cells / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_fused_cells.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if list(t.get("grid", [])) == [S, W] and bool(t.get("lazy_flux", False)) == lazy:
                traffic = t.get("hbm_bytes_per_launch")   # measured for this launch size and mode only
        except Exception:
            traffic = None
    # the kernel that carries the step is bound by VALU issue, not HBM: its measured share of busy
    # SIMD cycles comes from the committed PMC run (counters cannot be read from inside this process)
    proof = None
    ppath = os.path.join(ROOT, "profiles", "particle_roofline.json")
    if os.path.exists(ppath) and (S, W) == (8192, 8192):
        try:
            pr = json.load(open(ppath))
            proof = {"bound": pr["bound"], "kernel": "k_tiled_round", "achieved": pr["achieved"], "peak": 1.0,
                     "unit": pr["unit"], "frac": pr["achieved"], "measured_by": pr["source"],
                     "live": {"gparticle_steps_per_s": psteps_rank * world / elapsed / 1e9,
                              "particle_phase_ms": (phase[0] + phase[1]) / K}}
        except Exception:
            proof = None
    out = {
        "metric": "Mcells/s on %d^2 hydraulic-erosion step" % W,   # BASELINE.json's at the default size
        "value": cells / (elapsed / K) / 1e6,
        "unit": "Mcells/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "baseline_config": {"c4": "configs[3]" if world == 1 else "configs[3] weak-scaled", "c2": "configs[1]"}.get(args.config),
            "workload": "%dx%d coupled hydraulic+thermal erosion step (fluvial+debris particle "
                        "transport, N=cells/%d, maxage 256, + fused cell phase), OpenSimplex2-FBm "
                        "heightmap, example/erosion_gpu.py parameters%s" % (
                            H_global, W, args.particles_div,
                            "; + `strong16384`: the %d^2 grid of BASELINE configs[4] cut into %d row "
                            "slabs, same harness" % (G, world) if strong_block else ""),
            "grid": [H_global, W], "particles": cells // args.particles_div, "maxage": 256,
            "parallelism": "row-slabs x%d" % world if world > 1 else "single GPU",
            "particle_arithmetic": args.particle_arith + ": " + ARITH_NOTE[args.particle_arith],
            "arith_vs_reference": ("same as the reference's CUDA build: IEEE division and square root, __expf/__powf "
                                   "as the fast intrinsics the source names" if args.particle_arith == "exact" else
                                   "approximate reciprocal; reference is IEEE"),
            "debris_walkers": DEBRIS_NOTE[retire_mode],
        },
        "final_state": final,
        "halo": halo,
        "nccl_ranks": nccl_ranks,
        "particle_steps_per_step": psteps_rank * world // K,
        "gparticle_steps_per_s": psteps_rank * world / elapsed / 1e9,
        "phases_ms": ({"particles_fluvial": phase[0] / K, "particles_debris": phase[1] / K}
                      if serial else {"particles_fluvial+debris_overlapped": (phase[0] + phase[1]) / K}) | {
                      "cells_fused": phase[2] / K} | (
                      {"cells_fused": (phase[2] - phase[3] - phase[4]) / K,
                       "exchange_flux_exposed": phase[3] / K,
                       "exchange_field_exposed": phase[4] / K} if world > 1 else {}),
        "cell_phase_mcells_per_s": cells_rank / t_cells / 1e6 * world,
        "roofline": {"bound": "hbm", "kernel": "k_erode_cells_fused", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "stream_probe": probe,
                     "algorithmic_bytes_per_launch": cell_bytes * cells_rank,
                     "algorithmic_bytes_per_cell": cell_bytes,
                     "flux_planes": ("left to the next step's particle launches (their first rounds "
                                     "store instead of adding): 84 B/cell, the 28 B/cell re-zero of "
                                     "SURVEY 8d's 112-B floor is never written") if lazy else
                                    "re-zeroed by this kernel: SURVEY 8d's 112-B fused floor",
                     # what a plain read-modify-write stream reaches on THIS box against the same peak:
                     # the practical ceiling of a streaming kernel here (boxes of the pool read 0.69-0.73;
                     # the guide's measured copy is 0.79) — `frac` / `ceiling` says how close the kernel is
                     "ceiling": (probe["achieved"] / HBM_PEAK_GBS) if probe else None,
                     "frac_of_ceiling": (achieved / probe["achieved"]) if probe else None,
                     "avg_launch_ms": t_cells * 1e3},
    }
    out["roofline_particles"] = proof
    if other_block:
        out[other + "_arithmetic"] = other_block
    if walked_block:
        out["debris_walked_to_the_end"] = walked_block
    if strong_block:
        out["strong16384"] = strong_block
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_size)
    try:  # anything RCCL/HIP left in the C stdio buffer goes out BEFORE the result line
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def strong_scaling_block(args, weak_runner, ev, rank, world, param, G, comm, pair):
    """BASELINE.json configs[4] in the harness of the line above: the G x G grid cut into `world`
    row slabs (K timed steps, max over ranks), then the same grid on rank 0's GPU alone
    (`speedup_vs_1gpu` = the ratio of the two step times, both measured in this run)."""
    if G % world:
        return {"skipped": "%d rows do not split into %d equal slabs" % (G, world)}
    S = G // world
    if weak_runner.G > S:
        return {"skipped": "ghost depth %d exceeds the %d rows of a slab" % (weak_runner.G, S)}
    weak_runner.close()          # the weak run's planes go back to the device first
    runner = slab_runner(S, G, True, param, args.particles_div, comm, pair, args.halo_mode)
    K = args.steps
    elapsed, phase, psteps = timed_steps(runner, ev, K, args.warmup, world)
    block = {
        "grid": [G, G], "scaling": "strong", "n_gpus": world, "rows_per_gpu": S,
        "ms_per_step": elapsed / K * 1e3, "value": G * G / (elapsed / K) / 1e6, "unit": "Mcells/s",
        "steps": K, "warmup": args.warmup,
        "phases_ms": {"particles": (phase[0] + phase[1]) / K,
                      "cells_fused": (phase[2] - phase[3] - phase[4]) / K,
                      "exchange_flux_exposed": phase[3] / K,
                      "exchange_field_exposed": phase[4] / K},
        "halo": halo_report(runner),
        "gparticle_steps_per_s": psteps * world / elapsed / 1e9,
    }
    runner.close()
    # ... and the same split with the other way of serving walks that cross a slab's edge (soil_slab.h), so
    # that one multi-GPU run settles which of the two the wire favours
    other = "migrate" if args.halo_mode == "deep" else "deep"
    if os.environ.get("SOIL_BENCH_ONE_HALO_MODE") != "1" and not (other == "deep" and weak_runner.G > S):
        r2 = slab_runner(S, G, True, param, args.particles_div, comm, pair, other)
        e2, ph2, _ = timed_steps(r2, ev, K, args.warmup, world)
        block["other_halo_mode"] = {
            "mode": other, "ms_per_step": e2 / K * 1e3, "value": G * G / (e2 / K) / 1e6,
            "phases_ms": {"particles": (ph2[0] + ph2[1]) / K, "cells_fused": (ph2[2] - ph2[3] - ph2[4]) / K,
                          "exchange_flux_exposed": ph2[3] / K, "exchange_field_exposed": ph2[4] / K},
            "halo": halo_report(r2)}
        r2.close()
    # the same grid on one GPU (rank 0; the others wait at the barrier)
    if rank == 0 and os.environ.get("SOIL_BENCH_NO_1GPU_REF") != "1":
        k1 = max(2, min(K, 4))
        single = _Single(G, G, param, args.particles_div, serial=False)
        e1, _, _ = timed_steps(single, ev, k1, 1, 1)
        block["one_gpu"] = {"ms_per_step": e1 / k1 * 1e3, "value": G * G / (e1 / k1) / 1e6,
                            "steps": k1, "warmup": 1}
        block["speedup_vs_1gpu"] = (e1 / k1) / (elapsed / K)
        if "other_halo_mode" in block:
            block["other_halo_mode"]["speedup_vs_1gpu"] = (e1 / k1) / (block["other_halo_mode"]["ms_per_step"] * 1e-3)
        del single
    comm.barrier()
    return block


def _interleave(lib, _abi, layers, bed):
    """layers[..., 0] = bed, layers[..., 1] = 0, on the device (no host round trip)."""
    _abi.check(lib.soil_layers_from_planes(layers.c_ptr, bed.c_ptr, None, bed.elem(),
                                           _abi.stream()))


if __name__ == "__main__":
    main()
