"""GPU parity tests: every HIP entry point against the CPU oracle on the same
seeded inputs, through the C ABI.

Bars:
  * index / direction maps, and every deterministic per-cell kernel: BIT-EXACT
    (both sides evaluate the same fp32 expression trees with no contraction and
    share the software exp/pow and the Philox generator);
  * particle kernels: trajectories are bit-identical, only the ORDER of the fp32
    atomic additions differs -> flux planes compared with rtol 2e-5 of the
    per-cell accumulated magnitude (stated next to each check).
"""
import ctypes as C

import numpy as np
import pytest

from util import (assert_bit_equal, assert_receivers_close, product_param, rng_to_gpu, script_param, terrain,
                  to_gpu, to_np)

pytestmark = pytest.mark.gpu

D4, D8 = 0, 1
SIZES = [(64, 64), (37, 53), (128, 96)]       # square, ragged (W % 4 != 0), wide


def _selftest(hip, a, b, op):
    from soillib_amd import _abi
    ga, gb = to_gpu(a), to_gpu(b)
    out = to_gpu(np.zeros_like(a))
    _abi.check(hip.soil_selftest_math(out.c_ptr, ga.c_ptr, gb.c_ptr, a.size, op, None))
    return to_np(out)


# ----------------------------------------------------------- numerics

def test_device_is_gfx950(hip):
    buf = C.create_string_buffer(64)
    assert hip.soil_device_name(buf, 64) == 0
    assert buf.value.decode().startswith("gfx950"), buf.value


def test_spec_math_bit_exact(hip, oracle):
    r = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-100, 100, 40001), r.normal(0, 3, 20000),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 88.7, 88.73, -87.0, -87.1]]
                       ).astype(np.float32)
    assert_bit_equal(_selftest(hip, x, x, 0), oracle.expf(x), "expf_")
    assert_bit_equal(_selftest(hip, x, x, 7), oracle.expf(x), "expf_flat")
    xp = np.concatenate([np.exp(r.uniform(-80, 80, 30000)), [0.0, 1.0, 2.0, 1e-39, np.inf]]
                        ).astype(np.float32)
    assert_bit_equal(_selftest(hip, xp, xp, 1), oracle.log2f(xp), "log2f_")
    a = np.concatenate([r.uniform(0, 50, 30000), [0.0, 0.0, 1.0, 1e-30]]).astype(np.float32)
    b = np.concatenate([r.uniform(0.005, 3, 30000), [2.0, 0.0, 1.414, 0.01]]).astype(np.float32)
    assert_bit_equal(_selftest(hip, a, b, 2), oracle.powf(a, b), "powf_")


def test_particle_step_primitives(hip):
    """att_exp, floor_cell and sqrt_rn (soil_math.hpp), the three hardware-near pieces of the
    particle step, against what they stand for."""
    r = np.random.default_rng(3)
    # att_exp: the reference's __expf; only ever sees arguments <= 0 of modest size
    x = np.concatenate([-np.exp(r.uniform(-30, 4, 200000)), [0.0, -0.0, -87.0, -100.0, -1e30, np.nan]]
                       ).astype(np.float32)
    got = _selftest(hip, x, x, 8).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    ok = np.isfinite(want) & (want > 1e-37)
    # v_exp_f32 is good to 1 ulp; the product x * log2e carries half an ulp of |x| log2e into the
    # exponent, i.e. |x| 2^-24 into the result (the same holds for CUDA's __expf)
    err = np.abs(got[ok] / want[ok] - 1.0)
    assert (err <= 2.0 ** -22 + np.abs(x[ok].astype(np.float64)) * 1.5 * 2.0 ** -24).all()
    assert err[np.abs(x[ok]) < 1].max() < 2.0 ** -22               # where the attenuations live
    assert (got[~ok & ~np.isnan(want)] < 2e-37).all() and np.isnan(got[-1])
    # floor_cell: floor as int32, saturating, NaN -> INT_MAX (the kernels catch NaN themselves)
    f = np.concatenate([r.uniform(-5, 70000, 200000), [0.0, -0.0, -0.25, -1.0, -1.5, 8191.999, 8192.0,
                        16777215.0, 3e9, -3e9, np.inf, -np.inf, np.nan, 1e-45, -1e-45]]).astype(np.float32)
    got = _selftest(hip, f, f, 9).view(np.int32)
    want = np.floor(np.nan_to_num(f.astype(np.float64), nan=2.0 ** 31 - 1, posinf=2.0 ** 31 - 1,
                                  neginf=-2.0 ** 31))
    want = np.clip(want, -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
    assert (got.astype(np.int64) == want).all()
    # sqrt_rn: correctly rounded from 2^-96 up, and on +0, +inf, NaN
    e = r.integers(127 - 96, 255, 1 << 20).astype(np.uint32)
    m = r.integers(0, 1 << 23, 1 << 20, dtype=np.uint32)
    v = np.concatenate([((e << 23) | m).view(np.float32),
                        np.array([0.0, np.inf, np.nan, 2.0 ** -96, 1.0, 4.0, 2.0, 3.0], np.float32)])
    assert_bit_equal(_selftest(hip, v, v, 10), np.sqrt(v), "sqrt_rn vs numpy")
    assert_bit_equal(_selftest(hip, v, v, 11), np.sqrt(v), "device sqrtf vs numpy")
    tiny = (r.uniform(0, 1, 1000) * 2.0 ** -100).astype(np.float32)
    assert (_selftest(hip, tiny, tiny, 10) < 1e-12).all()          # all the step asks below 2^-96


def test_denormals_are_not_flushed(hip):
    a = np.array([1e-30, 3e-39, 1.5e-38], np.float32)
    b = np.array([1e-10, 0.5, 0.25], np.float32)
    assert_bit_equal(_selftest(hip, a, b, 4), a * b, "denormal product")


def test_shared_reciprocal_quotient_is_the_ieee_quotient(hip):
    """quot0(a, recip(b)) (soil_math.hpp) against the compiler's `/` on the device and
    numpy's on the host, over the plain range the particle step uses it on: random
    mantissas x every exponent combination, mantissa extremes, and +0 numerators."""
    r = np.random.default_rng(11)
    n = 1 << 22
    man = lambda k: r.integers(0, 1 << 23, k, dtype=np.uint32)
    def build(sign, exp, m):
        return ((sign.astype(np.uint32) << 31) | (exp.astype(np.uint32) << 23) | m).view(np.float32)
    for rep in range(4):
        eb = r.integers(127 - 40, 127 + 40, n)                 # |b| in [2^-40, 2^40)
        ea = r.integers(127 - 80, 127 + 50, n)                 # |a| in [2^-80, 2^50)
        a = build(r.integers(0, 2, n), ea, man(n))
        b = build(r.integers(0, 2, n), eb, man(n))
        if rep == 1:                                           # quotients next to 1: a ~ b
            a = (b.view(np.uint32) + r.integers(-3, 4, n).astype(np.uint32)).view(np.float32)
        if rep == 2:                                           # mantissa extremes
            edge = np.array([0, 1, 2, 0x7fffff, 0x7ffffe, 0x400000, 0x3fffff, 0x555555], np.uint32)
            a = build(r.integers(0, 2, n), ea, edge[r.integers(0, 8, n)])
            b = build(r.integers(0, 2, n), eb, edge[r.integers(0, 8, n)])
        if rep == 3:                                           # zeros of both signs in the mix
            a[::5] = 0.0
            a[1::5] = -0.0
        ieee = _selftest(hip, a, b, 5)
        fast = _selftest(hip, a, b, 6)
        assert_bit_equal(ieee, a / b, "device `/` vs numpy (rep %d)" % rep)
        assert_bit_equal(fast, ieee, "quot0 vs `/` (rep %d)" % rep)
    # any normal numerator: accepted by the size of the quotient (debris' decay_d)
    a = build(r.integers(0, 2, n), r.integers(1, 255, n), man(n))
    b = build(r.integers(0, 2, n), r.integers(127 - 40, 127 + 40, n), man(n))
    ieee, fast = _selftest(hip, a, b, 5), _selftest(hip, a, b, 6)
    keep = (np.abs(fast) >= 2.0 ** -60) & (np.abs(fast) <= 2.0 ** 90)
    assert keep.sum() > n // 4
    assert_bit_equal(fast[keep], ieee[keep], "quot0 vs `/`, accepted by the quotient")


def test_philox_uniform_bit_exact(hip, oracle):
    n = 5000
    seeds = np.full(n, 7, np.uint32).view(np.float32)
    offs = np.arange(n, dtype=np.uint32).view(np.float32)
    got = _selftest(hip, seeds, offs, 3)
    want = np.empty(n, np.float32)
    for i in range(n):
        st = oracle.rng_seed(1, 7, i)
        want[i] = oracle.rng_uniform(st, [i])[0]
    assert_bit_equal(got, want, "uniform")
    assert (got > 0).all() and (got <= 1).all()


# ----------------------------------------------------------- cell ops

def _cell_inputs(oracle, H, W, seed=0):
    r = np.random.default_rng(seed)
    layers = terrain(oracle, H, W, sediment=0.02, rng_seed=seed)
    f1 = lambda s: (r.random((H, W)) * s).astype(np.float32)
    f2 = lambda s: (r.standard_normal((H, W, 2)) * s).astype(np.float32)
    return dict(layers=layers, uplift=f1(1.0), rainfall=f1(2.0), waterFlux=f1(3.0),
                massFlux=f1(0.5), velocityFlux=f2(2.0), debrisFlux=f1(0.2),
                debrisVelocityFlux=f2(1.0))


@pytest.mark.parametrize("H,W", SIZES)
@pytest.mark.parametrize("which", ["default", "script"])
def test_mass_transfer_and_creep_bit_exact(hip, oracle, H, W, which):
    from soillib_amd import soil
    inp = _cell_inputs(oracle, H, W)
    op = oracle.default_param()
    if which == "script":
        script_param(op)
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    r = np.random.default_rng(3)
    mass = (r.random((H, W)) * 5).astype(np.float32)
    vel = (r.standard_normal((H, W, 2)) * 4).astype(np.float32)
    debris = (r.random((H, W)) * 2).astype(np.float32)
    delta0 = (r.standard_normal((H, W, 2)) * 0.01).astype(np.float32)

    want = delta0.copy()
    oracle.mass_transfer(want, inp["layers"], inp["uplift"], mass, vel, debris, None, None, None,
                         None, scale, op)
    oracle.mass_creep(want, inp["layers"], scale, op)

    g_delta, g_layers = to_gpu(delta0), to_gpu(inp["layers"])
    z1, z2 = to_gpu(np.zeros((H, W), np.float32)), to_gpu(np.zeros((H, W, 2), np.float32))
    soil.mass_transfer(g_delta, g_layers, to_gpu(inp["uplift"]), z1, to_gpu(mass), to_gpu(vel),
                       to_gpu(debris), z2, None, None, None, None, scale, pp)
    soil.mass_creep(g_delta, g_layers, scale, pp)
    assert_bit_equal(to_np(g_delta), want, "delta after mass_transfer+mass_creep")

    g_h = to_gpu(np.zeros((H, W), np.float32))
    soil.layer_merge(g_h, g_layers)
    assert_bit_equal(to_np(g_h), oracle.layer_merge(inp["layers"]), "layer_merge")


@pytest.mark.parametrize("H,W", [(48, 40)])
def test_mass_transfer_albedo_bit_exact(hip, oracle, H, W):
    from soillib_amd import soil
    inp = _cell_inputs(oracle, H, W)
    inp["layers"][::3, ::2, 1] = 0.0          # exercise the layer.y == 0 branch (erosion.cu:558)
    op = script_param(oracle.default_param())
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    r = np.random.default_rng(4)
    c3 = lambda: r.random((H, W, 3)).astype(np.float32) * 1.3
    mass, debris = inp["massFlux"] * 8, inp["debrisFlux"] * 30
    vel = inp["velocityFlux"]
    a_bed, a_fl, a_db, a_surf = c3(), c3(), c3(), c3()
    want_d = np.zeros((H, W, 2), np.float32)
    want_s = a_surf.copy()
    oracle.mass_transfer(want_d, inp["layers"], inp["uplift"], mass, vel, debris, a_bed, a_fl,
                         a_db, want_s, scale, op)
    g_d, g_s = to_gpu(np.zeros((H, W, 2), np.float32)), to_gpu(a_surf)
    z1, z2 = to_gpu(np.zeros((H, W), np.float32)), to_gpu(np.zeros((H, W, 2), np.float32))
    soil.mass_transfer(g_d, to_gpu(inp["layers"]), to_gpu(inp["uplift"]), z1, to_gpu(mass),
                       to_gpu(vel), to_gpu(debris), z2, to_gpu(a_bed), to_gpu(a_fl), to_gpu(a_db),
                       g_s, scale, pp)
    assert_bit_equal(to_np(g_d), want_d, "delta")
    assert_bit_equal(to_np(g_s), want_s, "albedo_surface")
    assert (want_s != a_surf).any()


@pytest.mark.parametrize("H,W", SIZES + [(256, 256), (8, 4), (1, 8), (5, 1)])
def test_fused_cells_bit_exact(hip, oracle, H, W):
    """soil_erode_cells_fused == normalize x2, transfer, creep, add, merge in sequence."""
    from soillib_amd import _abi
    inp = _cell_inputs(oracle, H, W, seed=H * 1000 + W)
    op = script_param(oracle.default_param())
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    want = oracle.erode_cells(inp["layers"], inp["uplift"], inp["rainfall"], inp["waterFlux"],
                              inp["massFlux"], inp["velocityFlux"], inp["debrisFlux"],
                              inp["debrisVelocityFlux"], scale, op)
    g = {k: to_gpu(v) for k, v in inp.items()}
    out1 = lambda: to_gpu(np.full((H, W), np.nan, np.float32))
    out2 = lambda: to_gpu(np.full((H, W, 2), np.nan, np.float32))
    g.update(layers_next=out2(), height=out1(), waterHeight=out1(), mass=out1(), velocity=out2(),
             debris=out1(), debrisVelocity=out2())
    planes = _abi.ErosionPlanes()
    for name in _abi._PLANES:
        setattr(planes, name, g[name].ptr)
    dom = _abi.Domain(H, W, 0, H, 0, H)
    _abi.check(hip.soil_erode_cells_fused(C.byref(planes), C.byref(dom), _abi.vec(scale, 3),
                                          pp._ref(), None))
    for name in ("layers_next", "height", "waterHeight", "mass", "velocity", "debris",
                 "debrisVelocity"):
        assert_bit_equal(to_np(g[name]), want[name], "fused " + name)
    for name in ("waterFlux", "massFlux", "velocityFlux", "debrisFlux", "debrisVelocityFlux"):
        assert (to_np(g[name]) == 0).all(), name + " not re-zeroed"
    assert_bit_equal(to_np(g["layers"]), inp["layers"], "input layers untouched")


def test_fused_cells_on_slabs_equals_whole_grid(hip, oracle):
    """Row slabs with one ghost row reproduce the single-domain result bit for bit."""
    from soillib_amd import _abi
    H, W = 96, 64
    inp = _cell_inputs(oracle, H, W, seed=9)
    op = script_param(oracle.default_param())
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    want = oracle.erode_cells(inp["layers"], inp["uplift"], inp["rainfall"], inp["waterFlux"],
                              inp["massFlux"], inp["velocityFlux"], inp["debrisFlux"],
                              inp["debrisVelocityFlux"], scale, op)
    got = {k: np.zeros_like(v) for k, v in want.items() if k != "delta"}
    for (o0, o1) in [(0, 40), (40, 41), (41, 96)]:              # owned global rows
        x0, x1 = max(0, o0 - 1), min(H, o1 + 1)
        rows = x1 - x0
        sl = slice(x0, x1)
        g = {k: to_gpu(v[sl]) for k, v in inp.items()}
        g.update(layers_next=to_gpu(np.zeros((rows, W, 2), np.float32)),
                 **{k: to_gpu(np.zeros((rows, W), np.float32))
                    for k in ("height", "waterHeight", "mass", "debris")},
                 **{k: to_gpu(np.zeros((rows, W, 2), np.float32))
                    for k in ("velocity", "debrisVelocity")})
        planes = _abi.ErosionPlanes()
        for name in _abi._PLANES:
            setattr(planes, name, g[name].ptr)
        dom = _abi.Domain(H, W, x0, rows, o0 - x0, o1 - x0)
        _abi.check(hip.soil_erode_cells_fused(C.byref(planes), C.byref(dom), _abi.vec(scale, 3),
                                              pp._ref(), None))
        for name in got:
            got[name][o0:o1] = to_np(g[name])[o0 - x0:o1 - x0]
    for name in got:
        assert_bit_equal(got[name], want[name], "slab " + name)
    # a slab whose first computed row lacks its ghost row is rejected
    bad = _abi.Domain(H, W, 10, 20, 0, 20)
    assert hip.soil_erode_cells_fused(C.byref(planes), C.byref(bad), _abi.vec(scale, 3),
                                      pp._ref(), None) == _abi.SOIL_ERR_INVALID_ARGUMENT


def test_albedo_ops_bit_exact(hip, oracle):
    from soillib_amd import soil
    H, W = 40, 56
    r = np.random.default_rng(11)
    layers = terrain(oracle, H, W, sediment=0.3)
    c3 = lambda: r.random((H, W, 3)).astype(np.float32)
    op = oracle.default_param()
    pp = product_param(op)
    scale = (1.0, 1.0, 80.0)
    uplift = r.random((H, W)).astype(np.float32)
    want = np.zeros((H, W, 3), np.float32)
    oracle.albedo_stratum(want, uplift, layers, scale, op, (0.9, 0.5, 0.1), (0.2, 0.3, 0.4), 5e4, 2.5)
    got = to_gpu(np.zeros((H, W, 3), np.float32))
    soil.albedo_stratum(got, to_gpu(uplift), to_gpu(layers), scale, pp, (0.9, 0.5, 0.1),
                        (0.2, 0.3, 0.4), 5e4, 2.5)
    assert_bit_equal(to_np(got), want, "albedo_stratum")
    assert len(np.unique(want[..., 0])) == 2

    bed, sed = c3(), c3()
    want = np.zeros((H, W, 3), np.float32)
    oracle.albedo_layer(want, bed, sed, layers, 12.0, (0.1, 0.2, 0.3))
    got = to_gpu(np.zeros((H, W, 3), np.float32))
    soil.albedo_layer(got, to_gpu(bed), to_gpu(sed), to_gpu(layers), 12.0, (0.1, 0.2, 0.3))
    assert_bit_equal(to_np(got), want, "albedo_layer")

    alb = c3()
    dis = (r.standard_normal((H, W)) * 40).astype(np.float32)
    want = alb.copy()
    oracle.albedo_discharge(want, dis, (0.1, 0.3, 0.8), 0.05, 0.7)
    got = to_gpu(alb)
    soil.albedo_discharge(got, to_gpu(dis), (0.1, 0.3, 0.8), 0.05, 0.7)
    assert_bit_equal(to_np(got), want, "albedo_discharge")


# ------------------------------------------------------- particle ops

def _flux_close(got, want, what):
    """Same trajectories, different fp32 summation order: compare against the
    magnitude accumulated in each cell.  The absolute term covers cells of the signed (velocity)
    planes whose deposits cancel: their rounding error goes with the magnitude of the terms — up
    to the plane's maximum — not with the sum that is left (a lost or doubled deposit would show
    as an error of the order of the cell's own value)."""
    scale = np.nanmax(np.abs(want)) + 1e-30
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=4e-6 * scale, err_msg=what)
    # the same set of visited cells.  Deposits that have decayed to the edge of the fp32 range are
    # exempt: the oracle's expf_ flushes below e^-87, the attenuations' v_exp_f32 (att_exp,
    # soil_math.hpp) below 2^-126 — a deposit of 1e-38 on one side, none on the other
    assert (got[np.abs(want) > 1e-25 * scale] != 0).all(), what + ": different set of visited cells"
    assert (np.abs(got[want == 0]) <= 1e-30 * scale).all(), what + ": different set of visited cells"


@pytest.fixture(params=["direct", "staged", "tiled", "tiled-full", "tiled-panels", "tiled-sparse", "tiled-sparse-full"])
def particle_mode(request, hip, monkeypatch):
    """The launch shapes of the particle kernels (soil_set_particle_mode).  "tiled-full": the tiled
    shape with the LDS-filling tiles (78 / 104 rows) that large grids get by default;
    "tiled-panels": with the queue scan of grids of more than 16384 tiles; "tiled-sparse": sparse tiles
    (fewer than 64 walkers) on the one-wave kernel with its hashed accumulators, however few they are."""
    if request.param == "tiled-full":
        monkeypatch.setenv("SOIL_TILED_SHAPE", "3")
    if request.param == "tiled-panels":
        monkeypatch.setenv("SOIL_TILED_PANELS", "1")
    if request.param.startswith("tiled-sparse"):   # every tile of under 64 walkers through the one-wave kernel, from round 1 on
        monkeypatch.setenv("SOIL_TILED_SPARSE", "1")
        monkeypatch.setenv("SOIL_TILED_SPARSE_MIN", "1")
        monkeypatch.setenv("SOIL_TILED_SPARSE_PCT", "1")
    if request.param == "tiled-sparse-full":       # ... whose table takes a cell only where its hash points:
        monkeypatch.setenv("SOIL_TILED_SPARSE_PROBE", "1")   # colliding cells add to the planes directly
    assert hip.soil_set_particle_mode(1 if request.param == "direct" else 2 if request.param == "staged" else 3) == 0
    yield request.param
    hip.soil_set_particle_mode(0)


@pytest.mark.parametrize("H,W,N", [(64, 64, 4096), (96, 40, 3000), (33, 50, 700), (200, 136, 20000)])
@pytest.mark.parametrize("which", ["default", "script"])
def test_transport_fluvial_parity(hip, oracle, particle_mode, H, W, N, which):
    from soillib_amd import soil
    op = oracle.default_param()
    if which == "script":
        script_param(op)
    op.maxage = 128
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    r = np.random.default_rng(21)
    layers = terrain(oracle, H, W, sediment=0.01)
    rain = (0.5 + r.random((H, W))).astype(np.float32)
    wh0 = (r.random((H, W)) * 0.1).astype(np.float32)
    vel0 = (r.standard_normal((H, W, 2)) * 2).astype(np.float32)
    asrc = r.random((H, W, 3)).astype(np.float32)
    z1, z2, z3 = (np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32),
                  np.zeros((H, W, 3), np.float32))

    o = dict(wh=wh0.copy(), wf=z1.copy(), m=z1.copy(), mf=z1.copy(), v=vel0.copy(), vf=z2.copy(),
             af=z3.copy())
    orng = oracle.rng_seed(N, 5, 100)
    steps = oracle.transport_fluvial(layers, rain, o["wh"], o["wf"], o["m"], o["mf"], o["v"],
                                     o["vf"], o["af"], asrc, orng, scale, op)
    assert steps > N            # the particles really move

    g = dict(wh=to_gpu(wh0), wf=to_gpu(z1), m=to_gpu(z1), mf=to_gpu(z1), v=to_gpu(vel0),
             vf=to_gpu(z2), af=to_gpu(z3))
    grng = rng_to_gpu(oracle.rng_seed(N, 5, 100))
    soil.particle_steps(reset=True)
    soil.transport_fluvial(to_gpu(layers), to_gpu(rain), g["wh"], g["wf"], g["m"], g["mf"], g["v"],
                           g["vf"], None, g["af"], to_gpu(asrc), grng, scale, pp)
    assert soil.particle_steps(reset=True) == steps     # same walks, step for step
    assert (to_np(grng)["offset"] == orng["offset"]).all()
    for k in ("wf", "mf", "vf"):
        _flux_close(to_np(g[k]), o[k], "fluvial flux " + k)
    for k in ("wh", "m", "v"):
        np.testing.assert_allclose(to_np(g[k]), o[k], rtol=3e-5,
                                   atol=3e-6 * (np.nanmax(np.abs(o[k])) + 1e-30), err_msg=k)
    np.testing.assert_allclose(to_np(g["af"]), o["af"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("H,W,N", [(64, 64, 4096), (40, 96, 3000), (50, 33, 700), (136, 200, 20000)])
def test_transport_debris_parity(hip, oracle, particle_mode, H, W, N):
    from soillib_amd import soil
    op = script_param(oracle.default_param())
    op.maxage = 128
    op.critSlopeBedrock = 0.05          # make landslides happen on the synthetic terrain
    op.yieldStress = 0.001
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    r = np.random.default_rng(22)
    layers = terrain(oracle, H, W, sediment=0.01)
    vel0 = (r.standard_normal((H, W, 2)) * 0.5).astype(np.float32)
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    asrc = r.random((H, W, 3)).astype(np.float32)          # colour of the debris source cells
    z3 = np.zeros((H, W, 3), np.float32)
    o = dict(v=vel0.copy(), vf=z2.copy(), m=z1.copy(), mf=z1.copy(), af=z3.copy())
    orng = oracle.rng_seed(N, 6, 0)
    steps = oracle.transport_debris(layers, o["v"], o["vf"], o["m"], o["mf"], o["af"], asrc, orng,
                                    scale, op)
    assert steps > N and o["mf"].max() > 0
    g = dict(v=to_gpu(vel0), vf=to_gpu(z2), m=to_gpu(z1), mf=to_gpu(z1), af=to_gpu(z3))
    grng = rng_to_gpu(oracle.rng_seed(N, 6, 0))
    soil.particle_steps(reset=True)
    soil.transport_debris(to_gpu(layers), g["v"], g["vf"], g["m"], g["mf"], None, g["af"],
                          to_gpu(asrc), grng, scale, pp)
    assert soil.particle_steps(reset=True) == steps
    np.testing.assert_allclose(to_np(g["af"]), o["af"], rtol=1e-3, atol=1e-5)
    for k in ("mf", "vf"):
        _flux_close(to_np(g[k]), o[k], "debris flux " + k)
    for k in ("m", "v"):
        np.testing.assert_allclose(to_np(g[k]), o[k], rtol=3e-5,
                                   atol=3e-6 * (np.nanmax(np.abs(o[k])) + 1e-30), err_msg=k)


def test_particles_on_slabs_equal_whole_grid(hip, oracle, particle_mode):
    """soil_particles_*_slab on three row slabs (ghost depth from soil_ghost_rows) deposit
    exactly what one launch on the whole grid deposits, NaN walkers included (remote0)."""
    from soillib_amd import _abi
    H, W, N = 96, 48, 6000
    op = script_param(oracle.default_param())
    op.maxage = 8
    op.critSlopeBedrock = 0.05
    pp = product_param(op)
    G = int(hip.soil_ghost_rows(pp._ref()))
    assert G == int(np.ceil(np.sqrt(2.0) * 8)) + 2
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers = terrain(oracle, H, W, sediment=0.01)
    rain = np.ones((H, W), np.float32)
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)

    def run(x0, rows, r0, r1):
        sl = slice(x0, x0 + rows)
        g = dict(wf=to_gpu(z1[sl]), mf=to_gpu(z1[sl]), vf=to_gpu(z2[sl]), df=to_gpu(z1[sl]),
                 dvf=to_gpu(z2[sl]), rem=to_gpu(np.zeros(8, np.float32)))
        rng = rng_to_gpu(oracle.rng_seed(N, 2, 0))
        dom = _abi.Domain(H, W, x0, rows, r0, r1)
        lay, rn, zz1, zz2 = to_gpu(layers[sl]), to_gpu(rain[sl]), to_gpu(z1[sl]), to_gpu(z2[sl])
        _abi.check(hip.soil_particles_fluvial_slab(
            g["wf"].c_ptr, g["mf"].c_ptr, g["vf"].c_ptr, None, rng.c_ptr, N, lay.c_ptr, rn.c_ptr,
            zz1.c_ptr, zz2.c_ptr, None, g["rem"].c_ptr, C.byref(dom), _abi.vec(scale, 3),
            pp._ref(), None))
        _abi.check(hip.soil_particles_debris_slab(
            g["df"].c_ptr, g["dvf"].c_ptr, None, rng.c_ptr, N, lay.c_ptr, zz2.c_ptr, None,
            g["rem"].c_ptr, C.byref(dom), _abi.vec(scale, 3), pp._ref(), None))
        assert (to_np(rng)["offset"] == 4).all()     # every stream advanced, owned or not
        return {k: to_np(v) for k, v in g.items()}

    whole = run(0, H, 0, H)
    assert (whole["rem"] == 0).all() and np.isnan(whole["wf"][0, 0])
    acc = {k: np.zeros_like(v, dtype=np.float64) for k, v in whole.items()}
    for (o0, o1) in [(0, 32), (32, 64), (64, 96)]:
        x0, x1 = max(0, o0 - G), min(H, o1 + G)
        part = run(x0, x1 - x0, o0 - x0, o1 - x0)
        for k in ("wf", "mf", "vf", "df", "dvf"):
            acc[k][x0:x1] += part[k]
        acc["rem"] += part["rem"]
    # the owner of global row 0 receives the parked NaN-walker deposits
    acc["wf"][0, 0] += acc["rem"][0]
    acc["mf"][0, 0] += acc["rem"][1]
    acc["vf"][0, 0] += acc["rem"][2:4]
    acc["df"][0, 0] += acc["rem"][4]
    acc["dvf"][0, 0] += acc["rem"][5:7]
    for k in ("wf", "mf", "vf", "df", "dvf"):
        np.testing.assert_allclose(acc[k], whole[k], rtol=2e-5,
                                   atol=2e-6 * (np.nanmax(np.abs(whole[k])) + 1e-30), err_msg=k)


def _close_but_for_stray_walks(got, want, rtol, atol, max_frac, what):
    """Free-running multi-step comparison: from the second step on the two sides walk on terrains
    that differ in the last bits (fp32 summation order of the deposits, the attenuations'
    exponential), and a particle that stands within that of a cell boundary or of the `speed < eps`
    exit goes another way — its deposits then sit in other cells.  All cells but a small fraction
    must agree within the tolerance."""
    bad = ~(np.abs(got - want) <= atol + rtol * np.abs(want))
    bad &= ~(np.isnan(got) & np.isnan(want))
    assert bad.mean() <= max_frac, "%s: %d of %d cells differ" % (what, bad.sum(), bad.size)


def test_erosion_model_fused_equals_unfused_and_oracle(hip, oracle):
    """Three whole steps: fused step == stand-alone-op step (bit-exact cell phase,
    atomics aside) == oracle composition."""
    from soillib_amd import silt
    from soillib_amd.erosion import ErosionModel
    H = W = 64
    N = H * W // 8
    op = script_param(oracle.default_param())
    op.maxage = 64
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers0 = terrain(oracle, H, W)

    def make():
        m = ErosionModel(H, W, scale, pp, N, seed=0)
        m.set_layers(to_gpu(layers0))
        silt.set(m.rainfall, 1.0)
        return m
    a, b = make(), make()

    # oracle state
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    st = dict(layers=layers0.copy(), wh=z1(), m=z1(), v=z2(), d=z1(), dv=z2())
    rain, uplift = np.ones((H, W), np.float32), z1()
    for step in range(3):
        a.step()
        b.step_unfused()
        rng = oracle.rng_seed(N, 0, step * N)
        wf, mf, vf, df, dvf = z1(), z1(), z2(), z1(), z2()
        oracle.particles_fluvial(wf, mf, vf, None, rng, st["layers"], rain, st["wh"], st["v"],
                                 None, scale, op)
        oracle.particles_debris(df, dvf, None, rng, st["layers"], st["dv"], None, scale, op)
        res = oracle.erode_cells(st["layers"], uplift, rain, wf, mf, vf, df, dvf, scale, op)
        st = dict(layers=res["layers_next"], wh=res["waterHeight"], m=res["mass"],
                  v=res["velocity"], d=res["debris"], dv=res["debrisVelocity"])
        for name, key in (("layers", "layers"), ("waterHeight", "wh"), ("velocity", "v"),
                          ("debrisVelocity", "dv"), ("mass", "m")):
            ga, gb = to_np(getattr(a, name)), to_np(getattr(b, name))
            tol = dict(rtol=1e-4, atol=1e-5 * (np.nanmax(np.abs(st[key])) + 1e-30))
            np.testing.assert_allclose(ga, gb, err_msg="fused vs unfused " + name, **tol)
            if step == 0:
                np.testing.assert_allclose(ga, st[key], err_msg="fused vs oracle " + name, **tol)
            else:
                _close_but_for_stray_walks(ga, st[key], tol["rtol"], tol["atol"], 5e-3,
                                           "fused vs oracle %s, step %d" % (name, step))
    assert np.abs(st["layers"] - layers0).max() > 0          # the terrain really eroded
    assert_bit_equal(to_np(a.height), to_np(a.layers)[..., 0] + to_np(a.layers)[..., 1], "height")


# ---------------------------------------------------------- graph ops

@pytest.mark.parametrize("H,W", SIZES)
@pytest.mark.parametrize("edge", [D4, D8])
def test_flow_maps_bit_exact(hip, oracle, H, W, edge):
    from soillib_amd import soil
    h = terrain(oracle, H, W)[..., 0].copy()
    h[3:6, 3:6] = h[4, 4]                       # a flat patch: no receiver there
    gh = to_gpu(h)
    assert_bit_equal(to_np(soil.steepest(gh, edge)), oracle.steepest(h, edge), "steepest")
    assert_bit_equal(to_np(soil.direction(gh, edge)), oracle.direction(h, edge), "direction")
    for off in (0, 7):   # (a tolerance on the weights, SURVEY 8 a9: counted, and only on CDF edges)
        assert_receivers_close(oracle, to_np(soil.random_weighted(gh, edge, 3, off, 10.0)),
                               oracle.random_weighted(h, edge, 3, off, 10.0), h, 4 if edge == D4 else 8, 3, off, 10.0)
    flow = oracle.steepest(h, edge)
    assert_bit_equal(to_np(soil.slope(gh, to_gpu(flow), (0.3, 0.7))),
                     oracle.slope(h, flow, (0.3, 0.7)), "slope")
    # any graph is a legal argument: receivers that are not neighbours (two rows down, across the
    # row's end, anywhere), no receiver, the cell itself
    r = np.random.default_rng(9)
    wild = r.integers(-1, H * W, size=(H, W)).astype(np.int32)
    idx = np.arange(H * W, dtype=np.int64).reshape(H, W)
    wild[::3, ::2] = np.clip(idx[::3, ::2] + 2 * W, 0, H * W - 1)
    wild[1::3, -1] = np.clip(idx[1::3, -1] + 1, 0, H * W - 1)        # wraps onto the next row
    wild[2::3, 0] = np.clip(idx[2::3, 0] + W - 1, 0, H * W - 1)      # "down-left" from column 0
    wild[5, 5] = idx[5, 5]
    assert_bit_equal(to_np(soil.slope(gh, to_gpu(wild), (0.3, 0.7))),
                     oracle.slope(h, wild, (0.3, 0.7)), "slope on an arbitrary graph")
    with pytest.raises(ValueError):
        soil.steepest(gh, 5)                    # invalid edge enumerator, graph.cu:88


@pytest.mark.parametrize("H,W", SIZES)
@pytest.mark.parametrize("edge", [D4, D8])
def test_accumulate_bit_exact(hip, oracle, H, W, edge):
    from soillib_amd import soil
    h = terrain(oracle, H, W)[..., 0].copy()
    r = np.random.default_rng(1)
    src = (0.5 + r.random((H, W))).astype(np.float32)
    for graph in (oracle.steepest(h, edge), oracle.random_weighted(h, edge, 0, 1, 10.0)):
        want = oracle.accumulate(graph, src, edge)
        got = to_np(soil.accumulate(to_gpu(graph), to_gpu(src), edge))
        assert_bit_equal(got, want, "accumulate")
        decay = (0.8 + 0.2 * r.random((H, W))).astype(np.float32)
        want = oracle.accumulate(graph, src, edge, decay=decay)
        got = to_np(soil.accumulate_decay(to_gpu(graph), to_gpu(src), to_gpu(decay), edge))
        assert_bit_equal(got, want, "accumulate_decay")
    ones = np.ones((H, W), np.float32)
    g = oracle.steepest(h, edge)
    acc = to_np(soil.accumulate(to_gpu(g), to_gpu(ones), edge))
    assert acc[g < 0].sum() == H * W            # every cell reaches exactly one outlet
    # `accumulate` is `accumulate_decay` with a decay of exactly 1 on every edge (graph.cu:577-583): the call
    # without a tensor runs without decay arrays (k_rake_compress<K, DECAY = false>), the one with an all-ones
    # tensor with them — the same floats in the same order either way
    for graph in (g, oracle.random_weighted(h, edge, 0, 2, 10.0)):
        a = to_np(soil.accumulate(to_gpu(graph), to_gpu(src), edge))
        b = to_np(soil.accumulate_decay(to_gpu(graph), to_gpu(src), to_gpu(ones), edge))
        assert_bit_equal(a, b, "accumulate against accumulate_decay with ones")


# ----------------------------------------------------------- stencils

@pytest.mark.parametrize("H,W", SIZES + [(1, 1), (2, 40)])
def test_stencils_bit_exact(hip, oracle, H, W):
    from soillib_amd import soil
    h = terrain(oracle, H, W)[..., 0].copy()
    gh = to_gpu(h)
    sc = (0.4, 1.7)
    assert_bit_equal(to_np(soil.gradient(gh, sc)), oracle.gradient(h, sc), "gradient")
    assert_bit_equal(to_np(soil.negslope(gh, sc)), oracle.negslope(h, sc), "negslope")
    r = np.random.default_rng(2)
    for D in (1, 2):
        t = r.standard_normal((H, W, D)).astype(np.float32)
        assert_bit_equal(to_np(soil.laplacian(to_gpu(t), sc)), oracle.laplacian(t, sc),
                         "laplacian D=%d" % D)
        gt = to_gpu(t)
        ret = soil.gaussian_blur(gt, 3.0)
        assert ret is gt                        # returns its input handle, filter.cu:90
        assert_bit_equal(to_np(gt), oracle.gaussian_blur(t, 3.0), "gaussian_blur C=%d" % D)
    s3 = (0.4, 1.7, 3.0)
    assert_bit_equal(to_np(soil.normal(gh, s3)), oracle.normal(h, s3), "normal (gpu)")


@pytest.mark.parametrize("H,W", [(70, 2052), (33, 1024), (100, 4), (65, 260)])
def test_window_kernels_across_seams(hip, oracle, H, W):
    """The four-cells-per-thread shape (window.hpp) where its seams are: rows wider than one
    work-group's 1024 columns, a wave's first and last lane (which reload their halo column),
    bands of 32 rows, a last group that ends the row, one group per row."""
    from soillib_amd import soil
    h = terrain(oracle, H, W)[..., 0].copy()
    if W > 262:
        h[5:9, 250:262] = h[6, 255]             # a flat patch across a wave boundary
    h[1, 1] = 1e-30                             # a height difference the shared-reciprocal quotient
    h[2, 1] = 3e-30                             # must hand to the written-out division
    if H > 40:
        h[40, 2] = np.inf
    # signed zeros: (-0) - (+0) = -0 is a numerator whose IEEE quotient is -0; the shared-reciprocal
    # chain would answer +0 (soil_math.hpp), so such groups must take the written-out division
    h[10:13, 1:3] = 0.0
    h[11, 1] = -0.0
    h[10, 2] = -0.0
    if W > 8:
        h[20:25, 4:9] = -0.0
        h[22, 6] = 0.0
    gh = to_gpu(h)
    sc = (0.4, 1.7)
    for edge in (D4, D8):
        assert_bit_equal(to_np(soil.steepest(gh, edge)), oracle.steepest(h, edge), "steepest")
        assert_bit_equal(to_np(soil.direction(gh, edge)), oracle.direction(h, edge), "direction")
    assert_bit_equal(to_np(soil.gradient(gh, sc)), oracle.gradient(h, sc), "gradient")
    assert_bit_equal(to_np(soil.negslope(gh, sc)), oracle.negslope(h, sc), "negslope")
    for D in (1, 2):
        t = np.random.default_rng(2).standard_normal((H, W, D)).astype(np.float32)
        assert_bit_equal(to_np(soil.laplacian(to_gpu(t), sc)), oracle.laplacian(t, sc),
                         "laplacian D=%d" % D)
    s3 = (0.4, 1.7, 3.0)                        # five rows / columns deep, inf and NaN samples
    h[3, 3] = np.nan
    assert_bit_equal(to_np(soil.normal(to_gpu(h), s3)), oracle.normal(h, s3), "normal")


@pytest.mark.parametrize("kind", ["quantised", "one_ulp", "two_ulp_binade", "tiny", "huge", "mixed"])
def test_steepest_ties_and_near_ties(hip, oracle, kind):
    """k_steepest4 takes one quotient per cell — that of the largest diagonal difference — and
    leaves the loop as written to the waves where a smaller difference could round to the same
    quotient (graph.hip).  Terrains made of exactly those cases: many equal differences (the first k
    must win), differences one and two bit patterns apart (the written loop must be taken: after the
    division by sqrt 2 they may or may not be equal), across a binade, and heights outside the plain
    range of the shared-reciprocal quotient (window.hpp)."""
    from soillib_amd import soil
    H, W = 96, 520                              # two waves per row and a ragged third
    r = np.random.default_rng(17)
    base = np.round(terrain(oracle, H, W)[..., 0] * 16.0) / 16.0   # plateaus, equal differences
    h = base.astype(np.float32)
    if kind != "quantised":
        steps = {"one_ulp": (1,), "two_ulp_binade": (1, 2, 3), "tiny": (1,), "huge": (1,), "mixed": (0, 1, 2, 4, 5)}[kind]
        if kind == "two_ulp_binade":
            h = (1.0 + np.round(r.random((H, W)) * 4.0) / 4.0).astype(np.float32)   # differences around 2^-2 .. 2^0
            h[::2, ::2] = 2.0
        bits = h.view(np.int32).copy()
        bump = r.choice(np.array(steps, np.int32), size=(H, W)) * (r.random((H, W)) < 0.5)
        h = (bits + bump.astype(np.int32)).view(np.float32)
        if kind == "tiny":
            h = (h * np.float32(2.0 ** -70)).astype(np.float32)     # differences below 2^-80: not plain
        if kind == "huge":
            h = (h * np.float32(2.0 ** 60)).astype(np.float32)
        if kind == "mixed":
            h[5:9, :] *= np.float32(2.0 ** -60)
            h[40, 100] = np.inf
            h[41, 300] = np.nan
            h[60:64, 250:262] = -0.0
    gh = to_gpu(np.ascontiguousarray(h))
    for edge in (D4, D8):
        assert_bit_equal(to_np(soil.steepest(gh, edge)), oracle.steepest(h, edge), "steepest, " + kind)
        assert_bit_equal(to_np(soil.direction(gh, edge)), oracle.direction(h, edge), "direction, " + kind)
    sc = (0.4, 1.7)
    assert_bit_equal(to_np(soil.negslope(gh, sc)), oracle.negslope(h, sc), "negslope, " + kind)
    big = (3e11, 2e-12)                         # scales at the ends of the plain range of a denominator
    assert_bit_equal(to_np(soil.negslope(gh, big)), oracle.negslope(h, big), "negslope, " + kind)


@pytest.mark.parametrize("H,W", [(70, 1500), (33, 1025), (300, 7), (300, 1028), (131, 260), (64, 2048)])
def test_gaussian_blur_across_tile_seams(hip, oracle, H, W):
    """Grids wider than one 1024-float LDS segment, taller than one 32-row band, row counts that
    are no multiple of the 8 rows a work-group of the axis-1 pass takes, widths with and without
    16-byte rows."""
    from soillib_amd import soil
    r = np.random.default_rng(H + W)
    for D in (1, 2):
        t = r.standard_normal((H, W, D)).astype(np.float32)
        gt = to_gpu(t)
        soil.gaussian_blur(gt, 2.5)
        assert_bit_equal(to_np(gt), oracle.gaussian_blur(t, 2.5), "gaussian_blur C=%d" % D)


def test_solve_uniform_parity(hip, oracle):
    from soillib_amd import soil
    H, W, N = 48, 40, 6000
    h = terrain(oracle, H, W)[..., 0].copy()
    sc = (0.05, 0.05)
    flow = -oracle.gradient(h, sc)
    r = np.random.default_rng(3)
    decay = (r.random((H, W)) * 0.01).astype(np.float32)
    for K in (1, 2):
        src = (r.random((H, W, K)) * 1e-3).astype(np.float32)
        want = oracle.solve_uniform(flow, src, decay, oracle.rng_seed(N, 1, 0), sc, N)
        got = to_np(soil.solve_uniform(to_gpu(flow), to_gpu(src), to_gpu(decay),
                                       rng_to_gpu(oracle.rng_seed(N, 1, 0)), sc, N))
        ok = np.isfinite(want)
        assert (np.isfinite(got) == ok).all()
        np.testing.assert_allclose(got[ok], want[ok], rtol=5e-5,
                                   atol=1e-6 * np.abs(want[ok]).max())


def test_noise_device_and_host_bit_exact(hip, oracle):
    from soillib_amd import silt, soil
    for (H, W, seed) in [(64, 64, 3.0), (50, 70, -1.5)]:
        p = soil.noise_t()
        p.seed = seed
        p.ext = [H, W]
        want = oracle.noise(H, W, seed=seed, ext=(H, W))
        assert_bit_equal(soil.noise(silt.shape(H, W), p).numpy(), want, "noise host")
        assert_bit_equal(to_np(soil.noise(silt.shape(H, W), p, host=silt.gpu)), want, "noise gpu")


def test_silt_gpu_ops(hip):
    from soillib_amd import _abi, silt
    a = np.arange(1003, dtype=np.float32)
    t = to_gpu(a)
    silt.multiply(t, 0.5)
    silt.add(t, to_gpu(np.ones(1003, np.float32)))
    np.testing.assert_array_equal(to_np(t), a * 0.5 + 1)
    c = silt.clone(t)
    silt.set(t, 3.0)
    assert (to_np(t) == 3).all() and (to_np(c) == a * 0.5 + 1).all()
    i = silt.tensor(silt.int32, silt.shape(7, 9), silt.gpu)
    silt.set(i, -1)
    assert (to_np(i) == -1).all()
    rg = silt.tensor(silt.rng, silt.shape(100), silt.gpu)
    silt.seed(rg, 42, 1000)
    got = to_np(rg)
    assert (got["seed"] == 42).all() and (got["offset"] == 1000).all()
    with pytest.raises(_abi.SoilError, match="mismatch_host"):
        from soillib_amd import soil
        soil.steepest(silt.tensor.from_numpy(np.zeros((4, 4), np.float32)), 0)


@pytest.mark.parametrize("old,new,D", [((16, 16), (32, 32), 1), ((24, 40), (61, 45), 2),
                                       ((33, 20), (33, 20), 1), ((40, 40), (17, 23), 3),
                                       ((5, 7), (1, 1), 1), ((1, 9), (4, 30), 2)])
def test_resize_bit_exact(hip, oracle, old, new, D):
    """soil.resize of the multiscale driver (soil_hip.h: soil_resize; parity unpinned —
    the reference snapshot has no definition, the oracle restates this build's)."""
    from soillib_amd import legacy, silt
    r = np.random.default_rng(31)
    src = r.standard_normal(old + ((D,) if D > 1 else ())).astype(np.float32)
    want = oracle.resize(src, new)
    dst = silt.tensor(silt.float32, silt.shape(*(new + ((D,) if D > 1 else ()))), silt.gpu)
    legacy.resize(dst, to_gpu(src), new, old)
    assert_bit_equal(to_np(dst), want, "resize")
    if old == new:
        assert_bit_equal(to_np(dst), src, "resize to the same resolution is the identity")
    const = np.full(old + ((D,) if D > 1 else ()), 3.25, np.float32)
    legacy.resize(dst, to_gpu(const), new, old)
    assert (to_np(dst) == 3.25).all()                 # weights sum to 1 exactly
    corners = to_np(legacy.resize(dst, to_gpu(src), new, old))
    if new[0] > 1 and new[1] > 1:
        assert_bit_equal(corners[0, 0], src[0, 0], "corner-aligned")
        assert_bit_equal(corners[-1, -1], src[-1, -1], "corner-aligned")


@pytest.mark.parametrize("H,W", [(64, 64), (97, 130), (200, 31), (1, 9), (300, 300)])
@pytest.mark.parametrize("edge", [0, 1])
def test_fill_depressions_bit_exact(hip, oracle, H, W, edge):
    """soil_fill_depressions vs the priority-flood oracle (build-defined):
    only min/max are involved, so the fixed point is exact whatever the update order."""
    from soillib_amd import soil
    r = np.random.default_rng(H * 7 + W + edge)
    dem = oracle.noise(H, W, seed=2.0, ext=(float(H), float(W))) * 50.0
    dem += (r.standard_normal((H, W)) * 2.0).astype(np.float32)        # plenty of pits
    if H > 8 and W > 8:
        dem[H // 3:H // 3 + 3, W // 2:W // 2 + 4] = np.nan                 # a NoData hole drains
        dem[H // 2, W // 4] = -1000.0                                     # a deep pit fills up
    want = oracle.fill_depressions(dem, edge)
    got = to_np(soil.fill_depressions(to_gpu(dem), edge))
    assert_bit_equal(got, want, "fill_depressions")
    ok = ~np.isnan(dem)
    assert (got[ok] >= dem[ok]).all() and np.isnan(got[~ok]).all()
    again = to_np(soil.fill_depressions(to_gpu(got), edge))
    assert_bit_equal(again, got, "filling is idempotent")
    if H > 8 and W > 8:
        assert got[H // 2, W // 4] > -1000.0
        # no cell is strictly below all of its neighbours any more (outlets aside)
        flow = to_np(soil.steepest(to_gpu(np.nan_to_num(got, nan=-1e9)), edge))
        inner = np.zeros((H, W), bool)
        inner[1:-1, 1:-1] = True
        pits = (flow < 0) & inner & ok
        # cells left without a receiver sit on flats (equal neighbours), never in a hole
        for x, y in zip(*np.nonzero(pits)):
            nb = got[x - 1:x + 2, y - 1:y + 2].copy()
            if edge == 0:
                nb[::2, ::2] = np.nan                      # D4: the diagonals are no neighbours
            assert np.nanmin(nb) >= got[x, y]


@pytest.mark.parametrize("H,W,maxage,tiled", [(512, 512, 96, False), (96, 96, 64, False),
                                                # widths that are a multiple of 4 but not of 256: the last
                                                # wave of a row of the four-cell pack pass has lanes past the
                                                # row's end (advisor finding of round 4: they stored their
                                                # share of the wave's records at another cell's address)
                                                (300, 520, 64, True), (200, 1000, 64, True), (130, 1928, 48, True),
                                                (260, 36, 48, True)])
def test_particle_pair_equals_sequential_launches(hip, H, W, maxage, tiled):
    """soil_particles_pair_slab (both launches overlapped on two streams, the debris launch
    on an rng tensor seeded two draws further) walks the same trajectories as the
    reference's order — fluvial, then debris on the same rng tensor.  The pair shares one pack
    pass (four cells per thread on widths that are a multiple of four), the sequential launches
    run the one-cell pass each: the cell records of the two are compared through the walks."""
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    param = script_param(soil.param_t())
    param.maxage = maxage
    param.critSlopeBedrock = 0.05
    param.yieldStress = 0.001
    if tiled:
        assert hip.soil_set_particle_mode(3) == 0
    try:
        _pair_equals_sequential(hip, H, W, param)
    finally:
        hip.soil_set_particle_mode(0)


def _pair_equals_sequential(hip, H, W, param):
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    m = ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), param, H * W // 8, seed=3)
    p = soil.noise_t()
    p.seed = 3.0
    p.ext = [H, W]
    bed = soil.noise(silt.shape(H, W), p, host=silt.gpu)
    zero = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
    silt.set(zero, 0.0)
    from soillib_amd import _abi
    _abi.check(hip.soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, zero.c_ptr, H * W, None))
    silt.set(m.rainfall, 1.0)
    m.step()
    m.step()
    # the step driver overlaps the launches too (scratch tensor for the fluvial draws): the caller's
    # rng tensor ends where the sequential order leaves it, four draws into step 1's streams
    assert (to_np(m.rng)["offset"] == 1 * m.N + 4).all()
    flux = ("waterFlux", "massFlux", "velocityFlux", "debrisFlux", "debrisVelocityFlux")
    out = {}
    for how in ("sequential", "pair"):
        for k in flux:
            silt.set(getattr(m, k), 0.0)
        m.seed_step()
        soil.particle_steps(reset=True)
        if how == "pair":
            m.particles_pair()
        else:
            m.particles_fluvial()
            m.particles_debris()
        steps = soil.particle_steps(reset=True)
        out[how] = (steps, {k: to_np(getattr(m, k)).copy() for k in flux},
                    to_np(m.rng_debris if how == "pair" else m.rng)["offset"].copy())
    assert out["pair"][0] == out["sequential"][0] > 0
    assert (out["pair"][2] == out["sequential"][2]).all()      # both end four draws further
    for k in flux:
        _flux_close(out["pair"][1][k], out["sequential"][1][k], "pair vs sequential " + k)


@pytest.mark.parametrize("maxage", [0, 1, 2, 33])
@pytest.mark.parametrize("N", [0, 1, 63, 5000])
def test_transport_edge_cases(hip, oracle, particle_mode, maxage, N):
    """Degenerate launches: no particles, one particle, lifetimes of 0 / 1 / 2 steps and one
    step past a round's budget — every launch shape against the oracle (fluvial and debris)."""
    from soillib_amd import soil
    H, W = 70, 90
    op = script_param(oracle.default_param())
    op.maxage = maxage
    op.critSlopeBedrock = 0.05
    op.yieldStress = 0.001
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 4.0)
    r = np.random.default_rng(5)
    layers = terrain(oracle, H, W, sediment=0.01)
    rain = np.ones((H, W), np.float32)
    wh0 = (r.random((H, W)) * 0.1).astype(np.float32)
    vel0 = (r.standard_normal((H, W, 2))).astype(np.float32)
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    o = dict(wf=z1.copy(), mf=z1.copy(), vf=z2.copy(), df=z1.copy(), dvf=z2.copy())
    orng = oracle.rng_seed(max(N, 1), 9, 40)[:N]
    steps = 0
    if N:
        steps += oracle.particles_fluvial(o["wf"], o["mf"], o["vf"], None, orng, layers, rain, wh0,
                                          vel0, None, scale, op)
        steps += oracle.particles_debris(o["df"], o["dvf"], None, orng, layers, vel0, None, scale, op)
    import ctypes as C
    from soillib_amd import _abi
    g = {k: to_gpu(v) for k, v in dict(wf=z1, mf=z1, vf=z2, df=z1, dvf=z2).items()}
    grng = rng_to_gpu(oracle.rng_seed(max(N, 1), 9, 40))
    lay, gr, gw, gv = to_gpu(layers), to_gpu(rain), to_gpu(wh0), to_gpu(vel0)
    dom = _abi.Domain(H, W, 0, H, 0, H)
    soil.particle_steps(reset=True)
    _abi.check(hip.soil_particles_fluvial_slab(
        g["wf"].c_ptr, g["mf"].c_ptr, g["vf"].c_ptr, None, grng.c_ptr if N else None, N, lay.c_ptr,
        gr.c_ptr, gw.c_ptr, gv.c_ptr, None, None, C.byref(dom), _abi.vec(scale, 3), pp._ref(), None))
    _abi.check(hip.soil_particles_debris_slab(
        g["df"].c_ptr, g["dvf"].c_ptr, None, grng.c_ptr if N else None, N, lay.c_ptr, gv.c_ptr,
        None, None, C.byref(dom), _abi.vec(scale, 3), pp._ref(), None))
    from util import debris_steps_match
    assert debris_steps_match(soil.particle_steps(reset=True), steps)     # (fluvial + debris)
    for k in ("wf", "mf", "vf", "df", "dvf"):
        _flux_close(to_np(g[k]), o[k], "edge case flux " + k)


@pytest.mark.parametrize("case", ["negative_shear", "tiny_cells", "huge_cells", "flat_axis", "force"])
def test_transport_fluvial_outside_the_plain_range(hip, oracle, case):
    """The tiled fluvial step takes its quotients from shared reciprocals when the launch
    constants and the lane's operands are plain (soil_math.hpp) and falls back to the
    written-out divisions otherwise.  Cases that sit on the other side of that switch —
    constants the fast path refuses, cell sizes far from 1, terrain whose gradient (hence
    a direction component) is exactly zero along one axis — against the oracle."""
    from soillib_amd import soil
    H, W, N = 96, 80, 6000
    op = script_param(oracle.default_param())
    op.maxage = 96
    scale = (20.0 / H, 20.0 / W, 4.0)
    r = np.random.default_rng(77)
    layers = terrain(oracle, H, W, sediment=0.01)
    vel0 = (r.standard_normal((H, W, 2)) * 2).astype(np.float32)
    if case == "negative_shear":
        op.bedShearWater = -0.5                    # tau + nu < 0: not plain, every lane slow
    elif case == "tiny_cells":
        scale = (1e-11, 1e-11, 4.0)                # lenL below 2^-30
    elif case == "huge_cells":
        scale = (3e9, 3e9, 4.0)                    # lenL above 2^30
    elif case == "flat_axis":                      # heights and speeds vary along x only
        layers = np.repeat(layers[:, :1], W, axis=1).copy()
        vel0[..., 1] = 0.0
    elif case == "force":
        op.force[0], op.force[1] = 0.3, -0.2
    pp = product_param(op)
    rain = np.ones((H, W), np.float32)
    wh0 = (r.random((H, W)) * 0.1).astype(np.float32)
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    o = dict(wf=z1.copy(), mf=z1.copy(), vf=z2.copy())
    orng = oracle.rng_seed(N, 3, 10)
    steps = oracle.particles_fluvial(o["wf"], o["mf"], o["vf"], None, orng, layers, rain, wh0, vel0,
                                     None, scale, op)
    import ctypes as C
    from soillib_amd import _abi
    out = {}
    for mode, name in ((1, "direct"), (3, "tiled")):
        assert hip.soil_set_particle_mode(mode) == 0
        g = {k: to_gpu(v) for k, v in dict(wf=z1, mf=z1, vf=z2).items()}
        grng = rng_to_gpu(oracle.rng_seed(N, 3, 10))
        dom = _abi.Domain(H, W, 0, H, 0, H)
        soil.particle_steps(reset=True)
        lay, gr, gw, gv = to_gpu(layers), to_gpu(rain), to_gpu(wh0), to_gpu(vel0)   # kept alive
        _abi.check(hip.soil_particles_fluvial_slab(
            g["wf"].c_ptr, g["mf"].c_ptr, g["vf"].c_ptr, None, grng.c_ptr, N, lay.c_ptr,
            gr.c_ptr, gw.c_ptr, gv.c_ptr, None, None, C.byref(dom),
            _abi.vec(scale, 3), pp._ref(), None))
        out[name] = (soil.particle_steps(reset=True), {k: to_np(v) for k, v in g.items()})
    hip.soil_set_particle_mode(0)
    for name, (n, flux) in out.items():
        assert n == steps, "%s: %d steps, oracle %d" % (name, n, steps)
        for k in ("wf", "mf", "vf"):
            _flux_close(flux[k], o[k], "%s %s flux %s" % (case, name, k))


def test_row_band_kernels_on_a_very_tall_grid(hip, oracle):
    """More than 65535 row bands (grid.y's limit): a work-group walks several bands."""
    from soillib_amd import soil
    H, W = 16 * 65535 + 123, 3
    r = np.random.default_rng(4)
    h = r.standard_normal((H, W)).astype(np.float32)
    gh = to_gpu(h)
    s3 = (0.4, 1.7, 3.0)
    assert_bit_equal(to_np(soil.normal(gh, s3)), oracle.normal(h, s3), "normal on a tall grid")
    flow = soil.steepest(gh, soil.d8)
    assert_bit_equal(to_np(flow), oracle.steepest(h, D8), "steepest on a tall grid")


@pytest.mark.parametrize("seed", range(8))
def test_transport_random_parameter_sets(hip, oracle, seed):
    """Fluvial and debris transport on the tiled shape under parameter sets drawn at random
    (log-uniform over several decades around the script's values, random cell sizes, an
    external force every other draw) against the oracle: same walks step for step, flux
    within the summation-order tolerance."""
    from soillib_amd import soil
    import ctypes as C
    from soillib_amd import _abi
    r = np.random.default_rng(1000 + seed)
    H, W, N = int(r.integers(70, 140)), int(r.integers(70, 140)), 6000
    op = script_param(oracle.default_param())
    op.maxage = int(r.integers(40, 130))
    lu = lambda lo, hi: float(np.exp(r.uniform(np.log(lo), np.log(hi))))
    op.gravity = lu(1.0, 30.0)
    op.evapRate = lu(1e-5, 1e-2)
    op.viscosityWater = lu(1e-7, 1e-2)
    op.bedShearWater = lu(0.05, 60.0)
    op.frictionFactor = lu(0.01, 1.0)
    op.depositionRateFluvial = lu(1e-7, 1e-2)
    op.suspensionRateFluvial = lu(1e-5, 1e-2)
    op.fluvialExponent = lu(0.01, 1.5)
    op.viscosityDebris = lu(1e-4, 0.1)
    op.bedShearDebris = lu(1e-3, 1.0)
    op.yieldStress = lu(1e-3, 1e7)
    op.critSlopeBedrock = lu(0.02, 0.8)
    op.landslideRateDebris = lu(1e-4, 1e-1)
    op.suspensionRateDebris = lu(1e-5, 1e-2)
    op.depositionRateDebris = lu(1e-5, 1e-2)
    if seed % 2:
        op.force[0], op.force[1] = float(r.normal(0, 0.3)), float(r.normal(0, 0.3))
    pp = product_param(op)
    scale = (lu(0.01, 3.0), lu(0.01, 3.0), lu(0.5, 8.0))
    layers = terrain(oracle, H, W, sediment=0.01, rng_seed=seed)
    rain = (0.5 + r.random((H, W))).astype(np.float32)
    wh0 = (r.random((H, W)) * lu(1e-3, 1.0)).astype(np.float32)
    vel0 = (r.standard_normal((H, W, 2)) * lu(0.01, 5.0)).astype(np.float32)
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    o = dict(wf=z1.copy(), mf=z1.copy(), vf=z2.copy(), df=z1.copy(), dvf=z2.copy())
    orng = oracle.rng_seed(N, 11, 5 * seed)
    steps_f = oracle.particles_fluvial(o["wf"], o["mf"], o["vf"], None, orng, layers, rain, wh0, vel0,
                                       None, scale, op)
    steps_d = oracle.particles_debris(o["df"], o["dvf"], None, orng, layers, vel0, None, scale, op)
    assert hip.soil_set_particle_mode(3) == 0
    try:
        g = {k: to_gpu(v) for k, v in dict(wf=z1, mf=z1, vf=z2, df=z1, dvf=z2).items()}
        grng = rng_to_gpu(oracle.rng_seed(N, 11, 5 * seed))
        lay, gr, gw, gv = to_gpu(layers), to_gpu(rain), to_gpu(wh0), to_gpu(vel0)
        dom = _abi.Domain(H, W, 0, H, 0, H)
        soil.particle_steps(reset=True)
        _abi.check(hip.soil_particles_fluvial_slab(
            g["wf"].c_ptr, g["mf"].c_ptr, g["vf"].c_ptr, None, grng.c_ptr, N, lay.c_ptr, gr.c_ptr,
            gw.c_ptr, gv.c_ptr, None, None, C.byref(dom), _abi.vec(scale, 3), pp._ref(), None))
        got_f = soil.particle_steps(reset=True)
        _abi.check(hip.soil_particles_debris_slab(
            g["df"].c_ptr, g["dvf"].c_ptr, None, grng.c_ptr, N, lay.c_ptr, gv.c_ptr, None, None,
            C.byref(dom), _abi.vec(scale, 3), pp._ref(), None))
        got_d = soil.particle_steps(reset=True)
    finally:
        hip.soil_set_particle_mode(0)
    from util import debris_steps_match
    assert got_f == steps_f and debris_steps_match(got_d, steps_d), (got_f, got_d, steps_f, steps_d)
    for k in ("wf", "mf", "vf", "df", "dvf"):
        _flux_close(to_np(g[k]), o[k], "random parameters, flux " + k)


@pytest.mark.parametrize("seed", range(6))
def test_fused_cells_random_parameter_sets_bit_exact(hip, oracle, seed):
    """The fused cell phase under parameter sets drawn at random (log-uniform over decades,
    random cell sizes, vector and ragged widths) against the oracle, bit for bit."""
    from soillib_amd import _abi
    r = np.random.default_rng(2000 + seed)
    H, W = int(r.integers(40, 120)), int(r.integers(10, 30)) * 4 + (seed % 2)   # W % 4 == 0 and != 0
    inp = _cell_inputs(oracle, H, W, seed=seed)
    op = script_param(oracle.default_param())
    lu = lambda lo, hi: float(np.exp(r.uniform(np.log(lo), np.log(hi))))
    op.timeStep = lu(1.0, 1e4)
    op.lrate = lu(0.01, 2.0)
    op.gravity = lu(1.0, 30.0)
    op.uplift = lu(1e-4, 1.0)
    op.rainfall = lu(0.01, 10.0)
    op.frictionFactor = lu(0.01, 1.0)
    op.fluvialExponent = lu(0.01, 1.5)
    op.suspensionRateFluvial = lu(1e-5, 1e-2)
    op.depositionRateFluvial = lu(1e-7, 1e-2)
    op.critSlopeBedrock = lu(0.02, 0.8)
    op.critSlopeSediment = lu(0.02, 0.8)
    op.landslideRateDebris = lu(1e-4, 1e-1)
    op.densityWater = lu(100.0, 2000.0)
    op.densityDebris = lu(500.0, 4000.0)
    pp = product_param(op)
    scale = (lu(0.01, 3.0), lu(0.01, 3.0), lu(0.5, 8.0))
    want = oracle.erode_cells(inp["layers"], inp["uplift"], inp["rainfall"], inp["waterFlux"],
                              inp["massFlux"], inp["velocityFlux"], inp["debrisFlux"],
                              inp["debrisVelocityFlux"], scale, op)
    g = {k: to_gpu(v) for k, v in inp.items()}
    out1 = lambda: to_gpu(np.full((H, W), np.nan, np.float32))
    out2 = lambda: to_gpu(np.full((H, W, 2), np.nan, np.float32))
    g.update(layers_next=out2(), height=out1(), waterHeight=out1(), mass=out1(), velocity=out2(),
             debris=out1(), debrisVelocity=out2())
    planes = _abi.ErosionPlanes()
    for name in _abi._PLANES:
        setattr(planes, name, g[name].ptr)
    dom = _abi.Domain(H, W, 0, H, 0, H)
    _abi.check(hip.soil_erode_cells_fused(C.byref(planes), C.byref(dom), _abi.vec(scale, 3),
                                          pp._ref(), None))
    for name in ("layers_next", "height", "waterHeight", "mass", "velocity", "debris",
                 "debrisVelocity"):
        assert_bit_equal(to_np(g[name]), want[name], "fused %s, random parameters" % name)


# ---------------------------------------------------------- deposits under contention

@pytest.mark.parametrize("knobs", [
    {"SOIL_TILED_AGG_MIN": "1", "SOIL_TILED_AGG_GROUPS": "1", "SOIL_TILED_RETRIES": "0"},
    {"SOIL_TILED_AGG_MIN": "1", "SOIL_TILED_AGG_GROUPS": "4", "SOIL_TILED_RETRIES": "1"},
    {"SOIL_TILED_AGG_MIN": "64", "SOIL_TILED_RETRIES": "2"},
    {"SOIL_TILED_AGG_MIN": "64", "SOIL_TILED_RETRIES": "0"},
    {"SOIL_TILED_DEP": "1"},
])
def test_contended_deposits_conserve_what_the_walkers_carry(hip, oracle, monkeypatch, knobs):
    """A funnel: every walker ends up in the same few channel cells, so nearly every lane of a wave
    loses its compare-and-swap.  Each way of making up for a lost swap — the wave-aggregated add
    (from one loser on), repeated swaps (0 / 1 / 2), the native add, and native adds only — must
    deposit exactly what the walkers carry.  The water plane is an exact counter here: parameters
    are chosen so that every deposit into it is exactly 1.0 (Q = 1/4, rain plane 4, no
    evaporation: att_w stays 1), and integers below 2^24 add up exactly in fp32 whatever the order
    — the plane must EQUAL the oracle's visit counts, one lost or doubled deposit anywhere shows.
    The other planes (paired with it in the 64-bit swaps) are compared within the summation-order
    tolerance."""
    from soillib_amd import soil
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    H = W = 128
    N = 65536
    x, y = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0] = 0.02 * np.hypot(x - 63.3, y - 64.7).astype(np.float32)   # a cone, apex inside a cell
    op = script_param(oracle.default_param())
    op.maxage = 96
    op.evapRate = 0.0
    op.rainfall = 1.0
    pp = product_param(op)
    scale = (1.0, 1.0, 4.0)                    # A = 1, Pr = 1 / 16384, Q = 1 / (Pr N) = 1/4, exactly
    rain = np.full((H, W), 4.0, np.float32)
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    o = dict(wh=z1.copy(), wf=z1.copy(), m=z1.copy(), mf=z1.copy(), v=z2.copy(), vf=z2.copy())
    orng = oracle.rng_seed(N, 3, 0)
    steps = oracle.particles_fluvial(o["wf"], o["mf"], o["vf"], None, orng, layers, rain, o["wh"], o["v"],
                                     None, scale, op)
    g = dict(wh=to_gpu(z1), wf=to_gpu(z1), m=to_gpu(z1), mf=to_gpu(z1), v=to_gpu(z2), vf=to_gpu(z2))
    assert hip.soil_set_particle_mode(3) == 0
    try:
        from soillib_amd import _abi
        soil.particle_steps(reset=True)
        _abi.check(hip.soil_particles_fluvial_slab(
            g["wf"].c_ptr, g["mf"].c_ptr, g["vf"].c_ptr, None, rng_to_gpu(oracle.rng_seed(N, 3, 0)).c_ptr,
            N, to_gpu(layers).c_ptr, to_gpu(rain).c_ptr, g["wh"].c_ptr, g["v"].c_ptr, None, None,
            C.byref(_abi.Domain(H, W, 0, H, 0, H)), _abi.vec(scale, 3), pp._ref(), None))
        assert soil.particle_steps(reset=True) == steps
    finally:
        hip.soil_set_particle_mode(0)
    got, want = {k: to_np(g[k]).copy() for k in ("wf", "mf", "vf")}, {k: o[k].copy() for k in ("wf", "mf", "vf")}
    for k in got:
        # cell (0,0) holds the NaN walkers' deposit on both sides (walkers spawned on the apex cell:
        # DESIGN.md, reference quirks); everything else is finite
        assert np.isnan(got[k][0, 0]).all() == np.isnan(want[k][0, 0]).all()
        got[k][0, 0] = 0.0
        want[k][0, 0] = 0.0
        assert np.isfinite(got[k]).all() and np.isfinite(want[k]).all()
    counts = want["wf"]
    assert (counts == np.round(counts)).all() and counts.max() < 2 ** 24
    assert counts.max() > 100 * np.median(counts[counts > 0])      # the funnel really is contended
    assert_bit_equal(got["wf"], counts, "visit counts (water plane)")
    assert counts.astype(np.float64).sum() > 0.5 * steps
    for k in ("mf", "vf"):
        _flux_close(got[k], want[k], "contended flux " + k)


# ---------------------------------------------------------- lazy flux planes

@pytest.mark.parametrize("S,maxage,mode", [(256, 64, 0), (512, 96, 0), (64, 32, 0), (256, 64, 1), (1024, 128, 0)])
def test_steps_without_rezeroing_the_flux_planes(hip, S, maxage, mode):
    """soil_erode_step_ex's lazy mode (what soil_erode and bench.py run): the cell phase leaves the
    flux planes as they are (SOIL_CELLS_KEEP_FLUX) and the next step's particle launches overwrite
    them (SOIL_FLUX_OVERWRITE) — by plain stores in their first round where every tile has exactly
    one work-group, by a clearing pass otherwise (small grids with empty or split tiles, the small-N
    launch shapes).  Every step must leave the same fields as the re-zeroing step does: the flux a
    cell receives is summed in the same order either way (0 + a == a), so the comparison is exact
    up to the order of the atomic adds of the late rounds (same tolerance as pair vs sequential)."""
    from soillib_amd import _abi, silt, soil
    from soillib_amd.erosion import ErosionModel
    pp = script_param(soil.param_t())
    pp.maxage = maxage
    scale = (20.0 / S, 20.0 / S, 4.0)
    npar = soil.noise_t()
    npar.seed = 3.0
    npar.ext = [S, S]
    bed = soil.noise(silt.shape(S, S), npar, host=silt.gpu)

    def make():
        m = ErosionModel(S, S, scale, pp, S * S // 8, seed=0)
        _abi.check(hip.soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, None, bed.elem(), None))
        silt.set(m.rainfall, 1.0)
        return m
    a, b = make(), make()
    assert hip.soil_set_particle_mode(mode) == 0
    try:
        dirty = False
        for step in range(4):
            a.seed_step()                     # eager: the reference's set(track.*, 0) after every step
            a.particles_pair()
            a.cells_fused()
            a.swap_layers()
            a.step_index += 1
            b.seed_step()                     # lazy
            b.particles_pair(overwrite=dirty)
            last = step == 3
            b.cells_fused(keep_flux=not last)
            dirty = not last
            b.swap_layers()
            b.step_index += 1
            for name in ("layers", "height", "waterHeight", "mass", "velocity", "debris", "debrisVelocity"):
                x, y = to_np(getattr(a, name)), to_np(getattr(b, name))
                tol = dict(rtol=1e-4, atol=1e-5 * (np.nanmax(np.abs(x)) + 1e-30))
                _close_but_for_stray_walks(y, x, tol["rtol"], tol["atol"], 0.0 if step == 0 else 2e-3,
                                           "step %d %s" % (step, name))
        for name in ("waterFlux", "massFlux", "velocityFlux", "debrisFlux", "debrisVelocityFlux"):
            assert not to_np(getattr(b, name)).any(), name + ": the last step of a chain re-zeroes"
    finally:
        hip.soil_set_particle_mode(0)


def test_erode_chain_leaves_the_track_planes_zeroed(hip):
    """soil_erode (the legacy composite) runs its steps lazily inside and hands the track planes
    back zeroed; its terrain equals a chain of plain soil_erode_step calls."""
    import soillib as soil
    from soillib_amd import silt
    S = 192
    q = soil.noise_t()
    q.seed = 3.0
    q.ext = [S, S]
    planes = lambda *dims: silt.tensor(silt.float32, silt.shape(*dims), silt.gpu)

    def run(chunks):
        model = soil.map_t(silt.shape(S, S), [20.0 / S, 20.0 / S, 4.0])
        model.height = soil.noise(silt.shape(S, S), q, host=silt.gpu)
        model.sediment, model.uplift, model.rainfall = planes(S, S), planes(S, S), planes(S, S)
        silt.set(model.sediment, 0.0)
        silt.set(model.uplift, 0.0)
        silt.set(model.rainfall, 1.0)
        data, track = soil.data_t(silt.shape(S, S)), soil.data_t(silt.shape(S, S))
        for d in (data, track):
            d.discharge, d.mass, d.debris = planes(S, S), planes(S, S), planes(S, S)
            d.momentum, d.debris_momentum = planes(S, S, 2), planes(S, S, 2)
            for t in (d.discharge, d.mass, d.debris, d.momentum, d.debris_momentum):
                silt.set(t, 0.0)
        p = soil.param_t()
        p.samples, p.maxage, p.timeStep = S * S // 8, 64, 1000.0
        for n in chunks:
            soil.erode(model, data, track, p, n)
        for t in (track.discharge, track.mass, track.debris, track.momentum, track.debris_momentum):
            assert not t.cpu().numpy().any()
        return model.height.cpu().numpy(), data.discharge.cpu().numpy()
    h4, d4 = run([4])            # one chain of four (steps 2..4 start from dirty planes)
    h1, d1 = run([1, 1, 1, 1])   # four chains of one (every step eager)
    _close_but_for_stray_walks(h4, h1, 1e-4, 1e-6, 2e-3, "height")
    _close_but_for_stray_walks(d4, d1, 1e-4, 1e-5 * np.nanmax(np.abs(d1)), 2e-3, "discharge")


@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_pair_modes_walk_the_same_walks(hip, monkeypatch, mode):
    """SOIL_PAIR_MODE: the two launches of a step overlapped freely (1), taking turns round by round
    behind the device-side gate (2: what grids of 4 M particles and more get), one after the other (3)
    — the same walks and the same fields as two separate launches."""
    from soillib_amd import _abi, silt, soil
    from soillib_amd.erosion import ErosionModel
    monkeypatch.setenv("SOIL_PAIR_MODE", mode)
    S = 512
    pp = script_param(soil.param_t())
    pp.maxage = 96
    scale = (20.0 / S, 20.0 / S, 4.0)
    npar = soil.noise_t()
    npar.seed = 3.0
    npar.ext = [S, S]
    bed = soil.noise(silt.shape(S, S), npar, host=silt.gpu)

    def make():
        m = ErosionModel(S, S, scale, pp, S * S // 8, seed=0)
        _abi.check(hip.soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, None, bed.elem(), None))
        silt.set(m.rainfall, 1.0)
        return m
    a, b = make(), make()
    for step in range(3):
        soil.particle_steps(reset=True)
        a.seed_step()
        a.particles_pair()
        sa = soil.particle_steps(reset=True)
        a.cells_fused()
        a.swap_layers()
        a.step_index += 1
        b.seed_step()
        b.particles_fluvial()
        b.particles_debris()
        sb = soil.particle_steps(reset=True)
        b.cells_fused()
        b.swap_layers()
        b.step_index += 1
        if step == 0:
            assert sa == sb
        for name in ("layers", "waterHeight", "velocity", "debris", "debrisVelocity"):
            x, y = to_np(getattr(a, name)), to_np(getattr(b, name))
            _close_but_for_stray_walks(x, y, 1e-4, 1e-5 * (np.nanmax(np.abs(y)) + 1e-30),
                                       0.0 if step == 0 else 2e-3, "step %d %s" % (step, name))


def test_two_host_threads_share_a_device(hip, oracle):
    """Two host threads step two models on one device at the same time (ctypes drops the GIL
    inside a call): scratch, streams and pinned words are per thread, so each gets the result it
    gets alone.  (With the per-device scratch of round 2 the second thread's spawn pass overwrote
    the first one's queues.)"""
    import threading
    from soillib_amd import erosion, silt, soil
    H = W = 1024
    param = script_param(soil.param_t())
    param.maxage = 128
    dem = terrain(oracle, H, W)

    def make():
        m = erosion.ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), param, H * W // 8, seed=3)
        m.set_layers(to_gpu(dem))
        silt.set(m.rainfall, 1.0)
        return m

    def run(m, out, key):
        for _ in range(3):
            m.step()
        out[key] = to_np(m.layers)

    alone = {}
    run(make(), alone, "ref")
    res = {}
    models = [make(), make()]
    threads = [threading.Thread(target=run, args=(m, res, i)) for i, m in enumerate(models)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        # trajectories are the same bits; the flux sums (fp32 atomics) are a tolerance between runs
        np.testing.assert_allclose(res[i], alone["ref"], rtol=2e-4, atol=1e-5)


def test_a_threads_scratch_goes_with_the_thread(hip, oracle):
    """A host thread's workspace blocks (hundreds of MB for the tiled particle launches) are freed when
    the thread ends: a caller that makes a thread per request does not pile them up (runtime.hip
    WsThreadGuard)."""
    import threading
    import torch
    from soillib_amd import erosion, silt, soil
    H = W = 2048
    param = script_param(soil.param_t())
    param.maxage = 64
    dem = terrain(oracle, 256, 256)
    dem = np.ascontiguousarray(np.kron(dem, np.ones((8, 8, 1), np.float32)))

    m = erosion.ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), param, H * W // 8, seed=3)
    m.set_layers(to_gpu(dem))
    silt.set(m.rainfall, 1.0)

    def run():
        m.step()
        hip.soil_device_synchronize()

    def free_now():
        hip.soil_device_synchronize()
        return torch.cuda.mem_get_info(0)[0]

    import time

    def settled(previous=None):
        # join() returns when the Python function has; the OS thread's thread_local destructors — where
        # the blocks go — run a moment later
        last = free_now()
        for _ in range(40):
            time.sleep(0.05)
            now = free_now()
            if now == last and (previous is None or previous - now < 64 * 2 ** 20):
                break
            last = now
        return last

    t = threading.Thread(target=run)        # (the first thread also pays one-off allocations of the runtime)
    t.start()
    t.join()
    base = settled()
    for _ in range(3):
        t = threading.Thread(target=run)
        t.start()
        t.join()
    after = settled(base)
    # one thread's scratch at this size is ~300 MB; three leaked sets would be ~0.9 GB
    assert base - after < 64 * 2 ** 20, "free memory fell by %d MB over three threads" % ((base - after) >> 20)

