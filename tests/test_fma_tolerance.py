"""The fp32 tolerance against the reference's own CUDA build, stated as numbers and tested.

nvcc compiles the reference with its default -fmad=true: every `a*b+c` the kernels spell out becomes
one fused multiply-add.  This build's contract is the unfused evaluation (-ffp-contract=off on both the
HIP and the oracle side), which is what makes bit-exact parity tests possible.  The only
statement about the reference's arithmetic that can still be made without nvcc is how far the two
evaluations of the SAME statements lie apart: oracle/liboracle_fma.so is the oracle's C restatement
compiled with -ffp-contract=fast -mfma (`pyoracle.contracted()`).

The bounds (`TOL`, quoted in DESIGN.md 4) are checked twice: oracle against contracted oracle on the
CPU (always), and the HIP kernels against the contracted oracle on the device (-m gpu).

* index / direction maps (steepest, direction, random_weighted on a shared draw): equal but for a
  counted handful of near-ties — none on the noise terrain;
* per-cell kernels: |difference| <= `cells_abs` x the plane's largest magnitude (a few ulp of the
  plane's scale; relative differences are unbounded where terms cancel);
* particle transport: trajectories are chaotic in the last bit, so the statement is statistical — the
  sums of the positive flux planes agree to `flux_sum_rel`, the sets of visited cells differ in at most
  `visited_frac` of the visited cells.
"""
import numpy as np
import pytest

from util import (product_param, rng_to_gpu, script_param, terrain, to_gpu, to_np)

TOL = {
    "cells_abs": 1e-6,        # per-cell kernels, relative to the plane's largest magnitude
    "index_frac": 1e-4,       # cells of an index map that may differ (near-ties)
    "flux_sum_rel": 2e-3,     # sums of the positive flux planes of one particle launch
    "visited_frac": 5e-3,     # cells visited on one side only
}

H = W = 192
SCALE = (20.0 / H, 20.0 / W, 4.0)


def _both(oracle, fn):
    plain = fn()
    with oracle.contracted():
        fused = fn()
    return plain, fused


def _close(got, want, what, tol=TOL["cells_abs"]):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    scale = np.nanmax(np.abs(want)) + 1e-30
    bad = ~(np.isclose(got, want, rtol=0.0, atol=tol * scale) | (np.isnan(got) & np.isnan(want)))
    assert not bad.any(), "%s: %d cells beyond %.1e of the plane's scale, worst %.3e" % (
        what, bad.sum(), tol, np.nanmax(np.abs(got - want)) / scale)


def _index_close(got, want, what):
    differ = int(np.sum(got != want))
    assert differ <= TOL["index_frac"] * got.size, "%s: %d of %d receivers differ" % (what, differ, got.size)


def _inputs(oracle):
    layers = terrain(oracle, H, W, sediment=0.02, rng_seed=5)
    return layers, np.ascontiguousarray(layers[..., 0] + layers[..., 1])


def _cell_inputs(oracle):
    layers, _ = _inputs(oracle)
    r = np.random.default_rng(11)
    f1 = lambda s: (r.random((H, W)) * s).astype(np.float32)
    f2 = lambda s: ((r.random((H, W, 2)) - 0.5) * s).astype(np.float32)
    p = script_param(oracle.default_param())
    return dict(layers=layers, uplift=f1(0.5), rain=np.ones((H, W), np.float32), wf=f1(300.0), mf=f1(3e-3),
                vf=f2(60.0), df=f1(1e-3), dvf=f2(0.5)), p


def _erode_cells(oracle, c, p):
    return oracle.erode_cells(c["layers"], c["uplift"], c["rain"], c["wf"], c["mf"], c["vf"], c["df"], c["dvf"],
                              SCALE, p)


def _particle_planes(oracle, threads=8):
    layers, _ = _inputs(oracle)
    p = script_param(oracle.default_param())
    p.maxage = 128
    N = H * W // 8
    r = np.random.default_rng(3)
    vel = ((r.random((H, W, 2)) - 0.5) * 0.1).astype(np.float32)
    wh = (r.random((H, W)) * 0.01).astype(np.float32)
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    out = dict(wf=z1(), mf=z1(), vf=z2())
    oracle.particles_fluvial(out["wf"], out["mf"], out["vf"], None, oracle.rng_seed(N, 0, 0), layers,
                             np.ones((H, W), np.float32), wh, vel, None, SCALE, p)
    return out, dict(layers=layers, vel=vel, wh=wh, p=p, N=N)


def _flux_statistics(got, want, what):
    for k in ("wf", "mf"):   # the positive planes: water and suspended mass carried
        a, b = np.nansum(got[k], dtype=np.float64), np.nansum(want[k], dtype=np.float64)
        assert abs(a - b) <= TOL["flux_sum_rel"] * abs(b), "%s %s: plane sums %.6e vs %.6e" % (what, k, a, b)
    for k in ("wf", "mf", "vf"):
        va, vb = got[k] != 0, want[k] != 0
        either = int(np.sum(va | vb))
        assert int(np.sum(va != vb)) <= TOL["visited_frac"] * either, "%s %s: %d of %d visited cells differ" % (
            what, k, int(np.sum(va != vb)), either)


# ---- the oracle against its contracted build (CPU) --------------------------------------------------

def test_stencils_contracted_vs_plain(oracle):
    _, h = _inputs(oracle)
    s2 = SCALE[:2]
    for name, fn in (("gradient", lambda: oracle.gradient(h, s2)), ("negslope", lambda: oracle.negslope(h, s2)),
                     ("laplacian", lambda: oracle.laplacian(h[..., None].copy(), s2)),
                     ("gaussian_blur", lambda: oracle.gaussian_blur(h[..., None].copy(), 2.0)),
                     ("normal", lambda: oracle.normal(h, SCALE))):
        a, b = _both(oracle, fn)
        _close(b, a, name)


def test_flow_graphs_contracted_vs_plain(oracle):
    _, h = _inputs(oracle)
    for name, fn in (("steepest", lambda: oracle.steepest(h, 8)), ("direction", lambda: oracle.direction(h, 8)),
                     ("random_weighted", lambda: oracle.random_weighted(h, 8, 0, 0, 10.0))):
        a, b = _both(oracle, fn)
        _index_close(b, a, name)
    g = oracle.steepest(h, 8)
    a, b = _both(oracle, lambda: oracle.slope(h, g, SCALE[:2]))
    _close(b, a, "slope")
    a, b = _both(oracle, lambda: oracle.accumulate(g, np.ones((H, W), np.float32), 8))
    _close(b, a, "accumulate")


def test_cell_phase_contracted_vs_plain(oracle):
    c, p = _cell_inputs(oracle)
    a, b = _both(oracle, lambda: _erode_cells(oracle, c, p))
    for k in a:
        _close(b[k], a[k], "cell phase " + k)


def test_particles_contracted_vs_plain(oracle):
    a, _ = _particle_planes(oracle)
    with oracle.contracted():
        b, _ = _particle_planes(oracle)
    _flux_statistics(b, a, "fluvial launch")


# ---- the HIP kernels against the contracted oracle (device) -----------------------------------------

@pytest.mark.gpu
def test_hip_stencils_within_the_stated_tolerance(hip, oracle):
    from soillib_amd import soil
    _, h = _inputs(oracle)
    s2 = SCALE[:2]
    gh = to_gpu(h)
    with oracle.contracted():
        want = dict(gradient=oracle.gradient(h, s2), negslope=oracle.negslope(h, s2),
                    laplacian=oracle.laplacian(h[..., None].copy(), s2)[..., 0],
                    blur=oracle.gaussian_blur(h[..., None].copy(), 2.0)[..., 0],
                    steepest=oracle.steepest(h, 8), direction=oracle.direction(h, 8),
                    rw=oracle.random_weighted(h, 8, 0, 0, 10.0))
        want["slope"] = oracle.slope(h, oracle.steepest(h, 8), s2)
    _close(to_np(soil.gradient(gh, s2)), want["gradient"], "gradient")
    _close(to_np(soil.negslope(gh, s2)), want["negslope"], "negslope")
    _close(to_np(soil.laplacian(to_gpu(h[..., None].copy()), s2))[..., 0], want["laplacian"], "laplacian")
    _close(to_np(soil.gaussian_blur(to_gpu(h[..., None].copy()), 2.0))[..., 0], want["blur"], "gaussian_blur")
    g = soil.steepest(gh, soil.d8)
    _index_close(to_np(g), want["steepest"], "steepest")
    _index_close(to_np(soil.direction(gh, soil.d8)), want["direction"], "direction")
    _index_close(to_np(soil.random_weighted(gh, soil.d8, 0, 0, 10.0)), want["rw"], "random_weighted")
    _close(to_np(soil.slope(gh, g, s2)), want["slope"], "slope")


@pytest.mark.gpu
def test_hip_cell_phase_within_the_stated_tolerance(hip, oracle):
    import ctypes as C
    from soillib_amd import _abi
    c, p = _cell_inputs(oracle)
    with oracle.contracted():
        want = _erode_cells(oracle, c, p)
    pp = product_param(p)
    out1 = lambda: to_gpu(np.full((H, W), np.nan, np.float32))
    out2 = lambda: to_gpu(np.full((H, W, 2), np.nan, np.float32))
    g = dict(layers=to_gpu(c["layers"]), layers_next=out2(), height=out1(), uplift=to_gpu(c["uplift"]),
             rainfall=to_gpu(c["rain"]), waterHeight=out1(), mass=out1(), velocity=out2(), debris=out1(),
             debrisVelocity=out2(), waterFlux=to_gpu(c["wf"]), massFlux=to_gpu(c["mf"]), velocityFlux=to_gpu(c["vf"]),
             debrisFlux=to_gpu(c["df"]), debrisVelocityFlux=to_gpu(c["dvf"]))
    planes = _abi.ErosionPlanes()
    for name in _abi._PLANES:
        setattr(planes, name, g[name].ptr)
    dom = _abi.Domain(H, W, 0, H, 0, H)
    _abi.check(hip.soil_erode_cells_fused(C.byref(planes), C.byref(dom), _abi.vec(SCALE, 3), pp._ref(), None))
    for k in ("layers_next", "height", "waterHeight", "mass", "velocity", "debris", "debrisVelocity"):
        _close(to_np(g[k]), want[k], "cell phase " + k)


@pytest.mark.gpu
def test_hip_particles_within_the_stated_tolerance(hip, oracle):
    from soillib_amd import soil
    with oracle.contracted():
        want, inp = _particle_planes(oracle)
    pp = product_param(inp["p"])
    z1 = lambda: to_gpu(np.zeros((H, W), np.float32))
    z2 = lambda: to_gpu(np.zeros((H, W, 2), np.float32))
    g = dict(wf=z1(), mf=z1(), vf=z2())
    assert hip.soil_set_particle_mode(3) == 0          # the tiled shape, whatever N
    try:
        soil.transport_fluvial(to_gpu(inp["layers"]), to_gpu(np.ones((H, W), np.float32)), to_gpu(inp["wh"]),
                               g["wf"], z1(), g["mf"], to_gpu(inp["vel"]), g["vf"], None, None, None,
                               rng_to_gpu(oracle.rng_seed(inp["N"], 0, 0)), SCALE, pp)
    finally:
        hip.soil_set_particle_mode(0)
    _flux_statistics({k: to_np(v) for k, v in g.items()}, want, "fluvial launch, HIP")
