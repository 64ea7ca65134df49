#!/usr/bin/env python
"""docs/KNOBS.md — every SOIL_* environment variable the product reads, generated from the call sites.

    python tools/gen_knobs.py            writes docs/KNOBS.md
    python tools/gen_knobs.py --check    exit 1 unless docs/KNOBS.md is what the sources say

The SITES come from the sources (every "SOIL_..." string literal handed to getenv / env_int / env_kind /
env_seconds / env_is / os.environ in soillib_amd/, bench.py, bench_configs.py, tests/conftest.py and
tests/cpp/watchdog.hpp); the DESCRIPTIONS live here, one line each.  A knob that is read somewhere and has no
description, or a description whose knob nobody reads any more, is an error — tests/test_knobs.py runs
--check, so the table cannot drift from the code.  (The `_F` / `_D` forms of the SOIL_TILED_* knobs: env_kind
looks for NAME_F / NAME_D, the fluvial / debris launch's own value, before NAME.)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["bench.py", "bench_configs.py", "tests/conftest.py", "tests/cpp/watchdog.hpp"]
for d in ("soillib_amd", "soillib_amd/csrc"):
    FILES += sorted(os.path.join(d, f) for f in os.listdir(os.path.join(ROOT, d))
                    if f.endswith((".py", ".hip", ".hpp")))

# area, meaning (defaults in brackets).  A/B = kept for measurements, not an interface.
KNOBS = {
    # ---- loading, devices, allocation
    "SOIL_LIB": ("load", "path of the C-ABI library to load instead of soillib_amd/lib/libsoil_hip.so (diagnostics builds: tools/prof_round.py)"),
    "SOIL_NO_TORCH": ("load", "1: do not import torch before loading the library (pure C-ABI use on the system ROCm; _abi.py)"),
    "SOIL_DEVICE": ("load", "HIP device of this process (default LOCAL_RANK, else 0): several ranks on one GPU in the tests"),
    "SOIL_POOL_BYTES": ("load", "limit of silt.py's stream-ordered pool of freed device blocks [16 GiB]"),
    # ---- the wire
    "SOIL_RCCL_LIB": ("wire", "librccl to dlopen instead of an already loaded copy / librccl.so.1 (slab_runner.hip rccl_load)"),
    "SOIL_RCCL_TIMEOUT_S": ("wire", "seconds a call into RCCL or a transfer on a stream may take before the communicator's watchdog aborts it (SOIL_ERR_COMM) [30; tests 20]"),
    "SOIL_RCCL_INIT_TIMEOUT_S": ("wire", "seconds ncclCommInitRank may take (a rank that never shows up) [120; tests 45]"),
    "SOIL_RCCL_WORLD1": ("wire", "1: a world of one still makes an RCCL communicator (default: the one-rank wire) — tests"),
    "SOIL_DIST_BACKEND": ("wire", "gloo: the ranks exchange through torch.distributed on staged host buffers (ranks sharing a GPU, CPU jobs) instead of RCCL"),
    # ---- slab runner
    "SOIL_SLAB_MODE": ("slabs", "deep | migrate: how a walk that crosses a slab edge is served when soil_slab_config.mode is -1 [deep]"),
    "SOIL_MIGRATE_HALO": ("slabs", "ghost rows a side in migrate mode [64]"),
    "SOIL_MIGRATE_PAIR": ("slabs", "0: the immigrants' launches kind by kind instead of both kinds side by side (A/B)"),
    "SOIL_HALO_FULL": ("slabs", "1: ship all ghost rows every step instead of the rows the measured reach asks for (A/B)"),
    "SOIL_HALO_NEED": ("slabs", "force the refresh depth of the field halo (rows); the tests use a depth that is too small to provoke repeated launches"),
    "SOIL_HALO_WINDOW": ("slabs", "0: every particle launch on all ghost rows instead of the fresh ones + 2 (A/B)"),
    "SOIL_SLAB_UNIFORM": ("slabs", "0/1: particle launches of a slab draw uniform streams (HipOps; A/B of round 4)"),
    "SOIL_SLAB_LAZY": ("slabs", "0: the slab's cell phase re-zeroes the flux planes itself instead of leaving them to the next launches' first rounds"),
    "SOIL_SLAB_VERBOSE": ("slabs", "set: per-pass timings of the migrate mode on stderr (synchronises)"),
    # ---- the step
    "SOIL_STEP_PAIR": ("step", "0: fluvial and debris launches one after the other instead of overlapped on two streams (read per step)"),
    "SOIL_PARTICLE_DIV": ("step", "fast: the process starts with the fast particle arithmetic (soil_set_particle_arith(1)) [exact]"),
    "SOIL_DEBRIS_RETIRE": ("step", "spent debris walkers — every further deposit certain to be an exact zero (soil_set_debris_retire): 1 end their walks, 0 are walked to the end as the reference walks them, 2 are marked, walked on and watched (the test suite's setting, tests/conftest.py) [1]"),
    # ---- tiled particle transport (erosion_particles_tiled.hip, TiledRun::setup); NAME_F / NAME_D per kind
    "SOIL_TILED_SHAPE": ("particles", "round kernel shape 0..3 = 64x64x512 | 64x64x768 | colour | LDS-filling 78/104 rows x 768 [by grid size]"),
    "SOIL_TILED_SHAPE_F": ("particles", "... of the fluvial launch"),
    "SOIL_TILED_SHAPE_D": ("particles", "... of the debris launch"),
    "SOIL_TILED_LATE": ("particles", "shape of the rounds from SOIL_TILED_SWITCH on [the early shape]"),
    "SOIL_TILED_SWITCH": ("particles", "round at which SOIL_TILED_LATE takes over [never]"),
    "SOIL_TILED_STEPS": ("particles", "steps a walker may take per round [fluvial 64 / 44 / 36, debris 40 / 32 / 48 by shape and residency]"),
    "SOIL_TILED_STEPS_LATE": ("particles", "... from round SOIL_TILED_LATE_FROM on [same]"),
    "SOIL_TILED_LATE_FROM": ("particles", "see SOIL_TILED_STEPS_LATE [never]"),
    "SOIL_TILED_TAIL": ("particles", "queue length below which the finishing launch takes over [N / 40 within 4096 .. 200000]"),
    "SOIL_TILED_FINISH_MRATE": ("particles", "M steps/s of the finishing launch the rounds' measured rate is compared with [4000]"),
    "SOIL_TILED_DEP": ("particles", "1: deposits as native ds_add_f32 instead of split-phase compare-and-swap (A/B; 64-row shapes only)"),
    "SOIL_TILED_RETRIES": ("particles", "swaps a lost deposit repeats before the native add [2 exact, 1 fast; 0 none]"),
    "SOIL_TILED_AGG_MIN": ("particles", "losers per wave from which their adds are combined per cell over DPP [48]"),
    "SOIL_TILED_AGG_GROUPS": ("particles", "distinct cells combined that way per iteration [4]"),
    "SOIL_TILED_STAGGER": ("particles", "1: odd rounds shift the tile grid by half a tile [fluvial on, debris off]"),
    "SOIL_TILED_SPARSE": ("particles", "1 / 2: force the one-wave kernel for tiles of < 64 walkers on / off [by grid size]"),
    "SOIL_TILED_SPARSE_MIN": ("particles", "sparse tiles a round must have for that kernel [64]"),
    "SOIL_TILED_SPARSE_PCT": ("particles", "... and their share of the non-empty tiles, per cent [25]"),
    "SOIL_TILED_SPARSE_PACK": ("particles", "2: tiles of < 32 walkers not packed two / four to a wave (A/B)"),
    "SOIL_TILED_SPARSE_PROBE": ("particles", "probes of the sparse kernel's hash table before a deposit goes to the planes [16]"),
    "SOIL_TILED_SLOTS": ("particles", "work-group slots of the chip the scan cuts long queues for [CUs x occupancy]"),
    "SOIL_TILED_AHEAD": ("particles", "rounds the host queues ahead of the scan words it has read [2; 0 with SOIL_TILED_VERBOSE]"),
    "SOIL_TILED_TAILSCAN": ("particles", "0: the queue scan as a launch of its own instead of in the round's last work-group (A/B)"),
    "SOIL_TILED_HOST_LAG_US": ("particles", "test hook: the host sleeps this long before it reads a scan word (a lagging host; test_step_at_1024_with_a_lagging_host)"),
    "SOIL_TILED_VERBOSE": ("particles", "set: per-round queue lengths and rates on stderr (synchronises every round)"),
    "SOIL_PAIR_MODE": ("particles", "how the two launches of a step share the chip: 0 taking turns through k_pair_gate, others: A/B variants"),
    "SOIL_PAIR_EARLY": ("particles", "per cent of a round's work-groups still to start when the other launch's next round is let in [20]"),
    "SOIL_PAIR_FREE": ("particles", "per cent of N below which the two launches stop taking turns [20]"),
    "SOIL_PAIR_DELAY": ("particles", "A/B: rounds the debris launch is held back at the start [0]"),
    "SOIL_PACK_PAIR": ("particles", "2: one pack pass per kind instead of the fused pass for both (A/B)"),
    "SOIL_PACK_WINDOW": ("particles", "2: the pack pass one cell per thread instead of the four-cell window (A/B)"),
    "SOIL_PACK_BAND": ("particles", "rows per band of the pack pass's window walk [kWinBand]"),
    "SOIL_ABLATE": ("particles", "-DSOIL_ABLATE builds only: bit mask of parts of the round kernel switched off (timing experiments; results wrong by construction)"),
    "SOIL_ABLATE_AFTER": ("particles", "... from the n-th launch of a kind on"),
    # ---- cells, stencils, graph
    "SOIL_CELLS_VARIANT": ("cells", "A/B variants of the fused cell kernel's launch (tools/bench_cells.py) [0]"),
    "SOIL_CELLS_NT": ("cells", "1: non-temporal accesses in the fused cell kernel (measured slower; A/B)"),
    "SOIL_CELLS_SPLIT": ("cells", "0: an eager step's cell phase as one 112-byte kernel instead of the 84-byte kernel + a zeroing pass"),
    "SOIL_WIN_SHAPE": ("stencils", "0..8: force the row-window walk of the four-cells-per-thread kernels (window.hpp) [per kernel and grid]"),
    "SOIL_WIN_FLAT_ORDER": ("stencils", "0: the flat window shape in natural block order instead of XCD-contiguous row ranges (A/B)"),
    "SOIL_LAP2_BAND": ("stencils", "rows per band of laplacian D=2 [16]"),
    "SOIL_NORMAL_BAND": ("stencils", "rows per band of the device normal kernel [8]"),
    "SOIL_RAKE_GROUPS": ("graph", "work-groups of a rake-compress round [8192]"),
    "SOIL_RAKE_LIST_FROM": ("graph", "first rake-compress round that runs over the lists of pending cells [2; 0: dense rounds throughout]"),
    "SOIL_FLOW_LANES": ("graph", "1: soil_multiflow's realisations one after the other instead of two in flight [2]"),
    "SOIL_FILL_PER_CHECK": ("graph", "relaxation launches of fill_depressions between two looks at the 'changed' word [3]"),
    "SOIL_FILL_FLAT": ("graph", "set: fill_depressions without the coarse levels (A/B)"),
    "SOIL_FILL_VERBOSE": ("graph", "set: launches per level on stderr"),
    # ---- bench.py
    "SOIL_BENCH_ARITH": ("bench", "default of --particle-arith [exact]"),
    "SOIL_BENCH_GRID": ("bench", "default of --grid (strong scaling on a fixed grid) [0]"),
    "SOIL_BENCH_STRONG_GRID": ("bench", "default of --strong-grid, the strong-scaling block a multi-GPU run appends [16384]"),
    "SOIL_BENCH_HALO_MODE": ("bench", "default of --halo-mode [deep]"),
    "SOIL_BENCH_ONE_HALO_MODE": ("bench", "1: the strong-scaling block without the other halo mode beside it"),
    "SOIL_BENCH_NO_1GPU_REF": ("bench", "1: the strong-scaling block without the whole grid on rank 0's GPU (speedup_vs_1gpu)"),
    "SOIL_BENCH_FORCE_SLAB": ("bench", "1: the N = 1 point through the slab runner"),
    "SOIL_BENCH_EAGER_FLUX": ("bench", "1: re-zero the flux planes after every step (the reference's set(track.*, 0)) instead of the lazy chain"),
    "SOIL_BENCH_NO_OTHER_ARITH": ("bench", "1: do not time the other particle arithmetic beside the headline (profiled runs)"),
    "SOIL_BENCH_NO_EXACT": ("bench", "round-5 name of SOIL_BENCH_NO_OTHER_ARITH, still honoured"),
    # ---- tests
    "SOIL_TEST_WATCHDOG_S": ("tests", "seconds after which a C++ test binary prints its last marker and ends itself (tests/cpp/watchdog.hpp) [60 / 75]"),
}
AREAS = [("load", "Loading, devices, allocation"), ("wire", "The wire between ranks (include/soil_slab.h)"),
         ("slabs", "Slab runner (csrc/slab_runner.hip)"), ("step", "The erosion step"),
         ("particles", "Tiled particle transport (csrc/erosion_particles_tiled.hip; NAME_F / NAME_D: per kind)"),
         ("cells", "Fused cell phase (csrc/erosion_cells.hip)"), ("stencils", "Stencil kernels (csrc/window.hpp, stencil.hip)"),
         ("graph", "Graph kernels and conditioning (csrc/graph.hip, conditioning.hip)"), ("bench", "bench.py"),
         ("tests", "Tests")]


def scan():
    sites = {}
    lit = re.compile(r'"(SOIL_[A-Z0-9_]+)"')
    ctx = re.compile(r"getenv|env_int|env_kind|env_seconds|env_is|environ")
    for rel in FILES:
        path = os.path.join(ROOT, rel)
        for no, line in enumerate(open(path, encoding="utf-8"), 1):
            if not ctx.search(line):
                continue
            for name in lit.findall(line):
                if name in ("SOIL_OK",) or name.startswith("SOIL_ERR"):
                    continue
                sites.setdefault(name, [])
                where = rel   # (the file, not the line: the table must not move with every edit)
                if where not in sites[name]:
                    sites[name].append(where)
    return sites


def render(sites):
    out = ["# SOIL_* environment variables", "",
           "Generated by `python tools/gen_knobs.py` from the call sites; `tests/test_knobs.py` fails when this file and",
           "the sources disagree.  Defaults in brackets.  \"A/B\": kept so that a measurement in DESIGN.md / docs/rounds can be",
           "repeated, not an interface.  The product's interface is the C ABI (`include/*.h`); nothing here is needed to use it.", ""]
    for area, title in AREAS:
        names = sorted(n for n in sites if KNOBS[n][0] == area)
        if not names:
            continue
        out += ["## " + title, "", "| variable | meaning | read at |", "|---|---|---|"]
        for n in names:
            out.append("| `%s` | %s | %s |" % (n, KNOBS[n][1], ", ".join("`%s`" % s for s in sites[n])))
        out.append("")
    return "\n".join(out)


def main():
    sites = scan()
    missing = sorted(n for n in sites if n not in KNOBS)
    stale = sorted(n for n in KNOBS if n not in sites)
    if missing or stale:
        print("tools/gen_knobs.py: read but not described: %s; described but not read: %s" % (missing, stale))
        return 1
    text = render(sites)
    path = os.path.join(ROOT, "docs", "KNOBS.md")
    if "--check" in sys.argv:
        if not os.path.exists(path) or open(path, encoding="utf-8").read() != text:
            print("docs/KNOBS.md is out of date: run python tools/gen_knobs.py")
            return 1
        return 0
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w", encoding="utf-8").write(text)
    print("wrote %s (%d variables)" % (path, len(sites)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
