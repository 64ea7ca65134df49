"""The default launch path (auto => tiled particle transport + fused cell phase) against the
oracle at BASELINE.json sizes: whole consecutive steps at 1024^2 (config C2's grid: N = 131 072
particles, maxage 256, example/erosion_gpu.py parameters) and on a 4096 x 512 strip, where the
tiled transport's split queues, staggered tile grid and finishing launch all come into play.

Two comparisons per step:
  * forced: the device starts the step from the ORACLE's state, so both sides walk on identical
    terrain — particle-step counts must be equal, flux planes agree to fp32 summation order,
    the cell phase's outputs to the tolerance that follows from that;
  * free-running: the device keeps its own state over all steps; from the second step on a few
    particles go another way (see test_gpu_parity._close_but_for_stray_walks), everything else
    must still agree.
Reference code matched: erosion.cu:189-239, :395-436 (transport), :576-757 (cell phase).
"""
import os

import numpy as np
import pytest

from util import product_param, script_param, terrain, to_gpu, to_np

pytestmark = pytest.mark.gpu

PLANES_IN = (("layers", "layers"), ("waterHeight", "wh"), ("velocity", "v"), ("debrisVelocity", "dv"))


def _oracle_step(oracle, st, step, N, scale, op, threads):
    H, W = st["layers"].shape[:2]
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    rain, uplift = np.ones((H, W), np.float32), z1()
    rng = oracle.rng_seed(N, 0, step * N)
    wf, mf, vf, df, dvf = z1(), z1(), z2(), z1(), z2()
    sf = oracle.particles_fluvial(wf, mf, vf, None, rng, st["layers"], rain, st["wh"], st["v"], None,
                                  scale, op, threads=threads)
    sd = oracle.particles_debris(df, dvf, None, rng, st["layers"], st["dv"], None, scale, op,
                                 threads=threads)
    res = oracle.erode_cells(st["layers"], uplift, rain, wf, mf, vf, df, dvf, scale, op)
    new = dict(layers=res["layers_next"], wh=res["waterHeight"], m=res["mass"], v=res["velocity"],
               d=res["debris"], dv=res["debrisVelocity"])
    return new, dict(wf=wf, mf=mf, vf=vf, df=df, dvf=dvf, steps_f=sf, steps_d=sd)


def _flux_close(got, want, what):
    """Same walks, another order of the fp32 additions into a cell.  At these sizes a channel cell
    collects 1e4..1e5 deposits per launch: the two orders then differ by a few 1e-5 of the sum
    (sqrt(n) * 2^-24 and up) — rtol 1e-4 here against 2e-5 in the small-grid tests."""
    scale = np.nanmax(np.abs(want)) + 1e-30
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6 * scale, err_msg=what)


def _run(hip, oracle, H, W, steps, warm_steps=0):
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    from test_gpu_parity import _close_but_for_stray_walks
    threads = os.cpu_count() or 1
    oracle.set_threads(threads)
    N = H * W // 8
    op = script_param(oracle.default_param())          # maxage 256, the script's parameters
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    assert hip.soil_set_particle_mode(0) == 0           # the default: auto => tiled at this N

    def make():
        m = ErosionModel(H, W, scale, pp, N, seed=0)
        silt.set(m.rainfall, 1.0)
        return m
    forced, free = make(), make()
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    st = dict(layers=terrain(oracle, H, W), wh=z1(), m=z1(), v=z2(), d=z1(), dv=z2())
    free.set_layers(to_gpu(st["layers"]))
    first = 0
    if warm_steps:   # let the device carve channels first; the comparison starts from its state
        for _ in range(warm_steps):
            free.step()
        first = warm_steps
        st = dict(layers=to_np(free.layers), wh=to_np(free.waterHeight), m=to_np(free.mass),
                  v=to_np(free.velocity), d=to_np(free.debris), dv=to_np(free.debrisVelocity))
    try:
        for step in range(first, first + steps):
            for name, key in PLANES_IN:                 # forced: this step starts from the oracle's state
                silt.set(getattr(forced, name), to_gpu(st[key]))
            forced.step_index = step
            forced.seed_step()
            soil.particle_steps(reset=True)
            forced.particles_fluvial()
            gsf = soil.particle_steps(reset=True)
            forced.particles_debris()
            gsd = soil.particle_steps(reset=True)
            flux = {k: to_np(getattr(forced, n)) for k, n in (
                ("wf", "waterFlux"), ("mf", "massFlux"), ("vf", "velocityFlux"), ("df", "debrisFlux"),
                ("dvf", "debrisVelocityFlux"))}
            forced.cells_fused()
            forced.swap_layers()
            free.step()                                  # soil_erode_step: the library's own driver
            st, o = _oracle_step(oracle, st, step, N, scale, op, threads)
            from util import debris_steps_match
            assert gsf == o["steps_f"] and debris_steps_match(gsd, o["steps_d"]), "step %d: particle steps" % step
            assert gsf > 20 * N                          # the walks are long ones
            for k in ("wf", "mf", "vf", "df", "dvf"):
                _flux_close(flux[k], o[k], "step %d flux %s" % (step, k))
            for name, key in (("layers", "layers"), ("waterHeight", "wh"), ("velocity", "v"),
                              ("debrisVelocity", "dv"), ("mass", "m"), ("debris", "d")):
                want = st[key]
                tol = dict(rtol=1e-4, atol=1e-5 * (np.nanmax(np.abs(want)) + 1e-30))
                _close_but_for_stray_walks(to_np(getattr(forced, name)), want, tol["rtol"], tol["atol"], 2e-6,
                                           "step %d forced %s" % (step, name))   # (same walks: see below)
                # (The first compared step walks the same trajectories on both sides; its fields still
                # differ by the order in which a cell's deposits were added up, which changes from run to run
                # — lost swaps, cut queues, the two launches overlapped.  Where a channel cell's deposits of
                # opposite sign cancel, that noise goes with the terms, not with the sum: one cell of 4.2 M
                # landed outside the tolerance in 2 of 40 runs of the 4096 x 512 strip.  A handful is allowed.)
                _close_but_for_stray_walks(to_np(getattr(free, name)), want, tol["rtol"], tol["atol"],
                                           2e-6 if step == first else 2e-3,
                                           "step %d free-running %s" % (step, name))
    finally:
        oracle.set_threads(1)
    assert np.abs(st["layers"] - terrain(oracle, H, W)).max() > 0     # the terrain really eroded


def test_three_steps_at_1024(hip, oracle):
    _run(hip, oracle, 1024, 1024, steps=3)


@pytest.mark.parametrize("lag_us", [200, 500, 1500])
def test_step_at_1024_with_a_lagging_host(hip, oracle, monkeypatch, lag_us):
    """The host queues rounds ahead of the queue words it has read.  A host that falls behind — here:
    that sleeps before every look at a word — finds the words of later scans written over the one it
    waits for, and must still size the rounds it queues next by the scan it asked about (round 3 took
    whatever count the word held: walkers dropped; found by eight processes sharing a GPU, where the
    hosts lag for real — tests/test_gpu_parallel.py)."""
    monkeypatch.setenv("SOIL_TILED_HOST_LAG_US", str(lag_us))
    _run(hip, oracle, 1024, 1024, steps=1)


def test_strip_4096x512_after_channels_formed(hip, oracle):
    _run(hip, oracle, 4096, 512, steps=2, warm_steps=3)


def test_4096x2048_default_launch_shapes(hip, oracle):
    """A grid large enough for the defaults of the big grids: LDS-filling tiles, 40-step fluvial
    rounds, one work-group per tile."""
    _run(hip, oracle, 4096, 2048, steps=1)


def test_1024_with_the_lds_filling_tiles(hip, oracle, monkeypatch):
    """The tile shape 8192^2 runs with by default (78 / 104 rows: not a power of two, queues
    longer than the work-group), forced onto the 1024^2 case."""
    monkeypatch.setenv("SOIL_TILED_SHAPE", "3")
    _run(hip, oracle, 1024, 1024, steps=2)
