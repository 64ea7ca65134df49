"""Spent debris walkers (include/soil_hip.h: soil_set_debris_retire; csrc/erosion_particles_tiled.hip:
debris_spent).  With the reference's example parameters a debris walker's attenuations underflow to exact
zeros within two steps and the reference walks it on for the rest of its life adding +-0
(/root/reference/source/soillib/model/path/erosion.cu:306-349); the product ends such a walk where that is
certain.  The rest of the suite runs in the WATCHED mode (tests/conftest.py): every walker walked to the end,
the marked ones' deposits counted — zero, after every test.  Here: the default mode against the walked-to-
the-end mode and the oracle, and the guards that switch the retirement off."""
import numpy as np
import pytest

from util import product_param, rng_to_gpu, script_param, terrain, to_gpu, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture
def retire(hip):
    """Sets the mode for one test; back to the suite's (watched) afterwards."""
    from soillib_amd import soil
    before = soil.debris_retire()
    yield soil.debris_retire
    soil.debris_retire(before)
    hip.soil_set_particle_mode(0)


def _same_planes(got, want, what, rtol=2e-5):
    """Same deposits, another order of the additions; the same cells visited."""
    scale = np.nanmax(np.abs(want)) + 1e-30
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all(), what + ": NaN cells differ"
    np.testing.assert_allclose(got[fin], want[fin], rtol=rtol, atol=4e-6 * scale, err_msg=what)
    assert ((got[fin] != 0) == (want[fin] != 0)).all() or \
        (np.abs(got[fin][(got[fin] != 0) != (want[fin] != 0)]) <= 1e-30 * scale).all(), what + ": visited cells differ"


def _debris_launch(soil, layers, vel0, N, scale, pp, seed=(6, 0)):
    from oracle import pyoracle
    H, W = layers.shape[:2]
    import ctypes as C
    from soillib_amd import _abi
    z1, z2 = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    g = dict(v=to_gpu(vel0), vf=to_gpu(z2), mf=to_gpu(z1))
    grng = rng_to_gpu(pyoracle.rng_seed(N, *seed))
    lay = to_gpu(layers)
    dom = _abi.Domain(H, W, 0, H, 0, H)
    soil.particle_steps(reset=True)
    _abi.check(_abi.lib().soil_particles_debris_slab(          # the launch alone, through the C ABI (soil_hip.h)
        g["mf"].c_ptr, g["vf"].c_ptr, None, grng.c_ptr, N, lay.c_ptr, g["v"].c_ptr,
        None, None, C.byref(dom), _abi.vec(scale, 3), pp._ref(), None))
    return soil.particle_steps(reset=True), to_np(g["mf"]), to_np(g["vf"])


@pytest.mark.parametrize("H,W,which", [(256, 256, "script"), (200, 136, "script"), (256, 192, "default"), (512, 512, "script")])
def test_retired_walks_leave_the_planes_of_the_full_walks(hip, oracle, retire, H, W, which):
    """On | off | the oracle: the same flux planes; fewer steps walked; nothing counted in the watched mode."""
    from soillib_amd import soil
    op = oracle.default_param()
    if which == "script":
        script_param(op)                       # example/erosion_gpu.py's parameters: yieldStress 2e6
    op.maxage = 128
    pp = product_param(op)
    N = H * W // 8
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers = terrain(oracle, H, W, sediment=0.01)
    vel0 = (np.random.default_rng(5).standard_normal((H, W, 2)) * 0.5).astype(np.float32)
    mf, vf = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    want_steps = oracle.particles_debris(mf, vf, None, oracle.rng_seed(N, 6, 0), layers, vel0, None, scale, op)
    assert hip.soil_set_particle_mode(3) == 0   # the tiled shape whatever N
    out = {}
    for mode in ("off", "watch", "on"):
        retire(mode)
        out[mode] = _debris_launch(soil, layers, vel0, N, scale, pp)
    assert soil.debris_retire_violations(reset=True) == 0
    assert out["off"][0] == want_steps and out["watch"][0] == want_steps      # the reference's walks, step for step
    assert out["on"][0] <= want_steps
    if which == "script":
        assert out["on"][0] < want_steps // 3, "the example's parameters: walkers are spent within a few steps"
    for mode in ("off", "on"):
        _same_planes(out[mode][1], mf, "%s: debris mass flux against the oracle" % mode)
        _same_planes(out[mode][2], vf, "%s: debris velocity flux against the oracle" % mode)
    _same_planes(out["on"][1], out["off"][1], "debris mass flux, on against off")
    _same_planes(out["on"][2], out["off"][2], "debris velocity flux, on against off")


@pytest.mark.parametrize("seed", range(6))
def test_random_parameter_sets_watched(hip, oracle, retire, seed):
    """Random launch constants and terrains: whoever gets marked never adds anything again; on == off."""
    from soillib_amd import soil
    r = np.random.default_rng(100 + seed)
    op = script_param(oracle.default_param())
    op.maxage = int(r.integers(40, 200))
    op.critSlopeBedrock = float(r.choice([0.01, 0.2, 0.57, 2.0]))
    op.yieldStress = float(r.choice([0.0, 1e-6, 1e-3, 1.0, 2e6]))
    op.landslideRateDebris = float(r.choice([0.0, 0.0025, 0.5]))
    op.depositionRateDebris = float(r.choice([0.0, 1e-4, 0.1]))
    op.suspensionRateDebris = float(r.choice([0.0, 2.5e-4, 0.1]))
    op.viscosityDebris = float(r.choice([0.0, 0.004, 1.0]))
    op.bedShearDebris = float(r.choice([0.0, 0.024, 0.99]))
    pp = product_param(op)
    H, W = int(r.integers(64, 200)), int(r.integers(64, 200))
    N = H * W // int(r.choice([2, 8]))
    zs = float(r.choice([0.5, 4.0, 40.0]))
    scale = (20.0 / H, 20.0 / W, zs)
    layers = terrain(oracle, H, W, sediment=float(r.choice([0.0, 0.01])), rng_seed=seed)
    vel0 = (r.standard_normal((H, W, 2)) * float(r.choice([0.0, 0.5, 5.0]))).astype(np.float32)
    assert hip.soil_set_particle_mode(3) == 0
    retire("watch")
    steps_w, mf_w, vf_w = _debris_launch(soil, layers, vel0, N, scale, pp)
    assert soil.debris_retire_violations(reset=True) == 0
    retire("on")
    steps_on, mf_on, vf_on = _debris_launch(soil, layers, vel0, N, scale, pp)
    assert steps_on <= steps_w
    _same_planes(mf_on, mf_w, "mass flux")
    _same_planes(vf_on, vf_w, "velocity flux")


@pytest.mark.parametrize("case", ["nan cell", "inf cell", "negative viscosity", "huge bed shear", "no yield stress on a cliff"])
def test_guards_switch_the_retirement_off(hip, oracle, retire, case):
    """Where the argument does not hold — a cell whose record is not finite, an excess stress that is not
    negative, launch constants out of range — nobody retires: the default mode walks the oracle's steps."""
    from soillib_amd import soil
    H = W = 128
    N = H * W // 4
    op = script_param(oracle.default_param())
    op.maxage = 64
    layers = terrain(oracle, H, W, sediment=0.0)
    scale = (20.0 / H, 20.0 / W, 4.0)
    if case == "nan cell":
        layers[70, 31, 0] = np.nan
    elif case == "inf cell":
        layers[20, 90, 0] = np.inf
    elif case == "negative viscosity":
        op.viscosityDebris = -0.004
    elif case == "huge bed shear":
        op.bedShearDebris = 1e30
    else:
        op.yieldStress = 0.0                    # excessStress = g * excessSlope: positive on the cliff below
        layers[64:, :, 0] += 50.0
    pp = product_param(op)
    vel0 = np.zeros((H, W, 2), np.float32)
    mf, vf = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    want = oracle.particles_debris(mf, vf, None, oracle.rng_seed(N, 6, 0), layers, vel0, None, scale, op)
    assert hip.soil_set_particle_mode(3) == 0
    retire("on")
    steps, got_mf, got_vf = _debris_launch(soil, layers, vel0, N, scale, pp)
    if case == "nan cell":
        # (a NaN height is the reference's sentinel for "outside the grid", erosion_map.cu:122-125: the slopes
        # next to it are the exit slope, every record stays finite and the argument holds — same planes below)
        assert steps < want
    else:
        assert steps == want, case
    _same_planes(got_mf, mf, case + ": mass flux", rtol=1e-4)
    _same_planes(got_vf, vf, case + ": velocity flux", rtol=1e-4)


def test_whole_steps_on_against_off(hip, oracle, retire):
    """Steps of the model's own driver (overlapped launches, fused cell phase) at 1024^2 in either mode: the
    same fields after the first step; after three, all but the stray walks of a free-running comparison
    (tests/test_gpu_parity.py: _close_but_for_stray_walks — the two runs' terrains differ in the last bits from
    the second step on, whatever the mode: the order of the fp32 additions into a cell changes from run to run)."""
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    from test_gpu_parity import _close_but_for_stray_walks
    H = W = 1024
    pp = product_param(script_param(oracle.default_param()))
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers = terrain(oracle, H, W)
    names = ("layers", "waterHeight", "velocity", "debris", "debrisVelocity", "mass")
    out = {}
    for mode in ("off", "on"):
        retire(mode)
        m = ErosionModel(H, W, scale, pp, H * W // 8, seed=0)
        m.set_layers(to_gpu(layers))
        silt.set(m.rainfall, 1.0)
        soil.particle_steps(reset=True)
        m.step()
        first = {n: to_np(getattr(m, n)) for n in names}
        m.step()
        m.step()
        out[mode] = (soil.particle_steps(reset=True), first, {n: to_np(getattr(m, n)) for n in names})
    assert out["on"][0] < out["off"][0]
    for n in names:
        want = out["off"][1][n]
        _close_but_for_stray_walks(out["on"][1][n], want, 1e-4, 1e-5 * (np.nanmax(np.abs(want)) + 1e-30), 2e-6, "step 1 " + n)
        want = out["off"][2][n]
        _close_but_for_stray_walks(out["on"][2][n], want, 1e-4, 1e-5 * (np.nanmax(np.abs(want)) + 1e-30), 2e-3, "step 3 " + n)


def test_bench_workload_8192(hip, oracle, retire):
    """BASELINE config 4's debris launch: the retired launch leaves the planes of the full one."""
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    import bench
    S = 8192
    pp = bench.script_param(soil)
    out = {}
    for mode in ("off", "on"):
        retire(mode)
        m = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), pp, S * S // 8, seed=0)
        npar = soil.noise_t()
        npar.seed = 3.0
        npar.ext = [S, S]
        from soillib_amd import _abi
        bench._interleave(hip, _abi, m.layers, soil.noise(silt.shape(S, S), npar, host=silt.gpu))
        silt.set(m.rainfall, 1.0)
        m.seed_step()
        soil.particle_steps(reset=True)
        m.particles_debris()
        out[mode] = (soil.particle_steps(reset=True), to_np(m.debrisFlux), to_np(m.debrisVelocityFlux))
        del m
    assert out["on"][0] * 5 < out["off"][0]
    _same_planes(out["on"][1], out["off"][1], "8192^2 debris mass flux")
    _same_planes(out["on"][2], out["off"][2], "8192^2 debris velocity flux")


@pytest.mark.parametrize("pair", [False, True])
def test_row_slabs_in_the_default_mode(hip, oracle, retire, pair):
    """The library's slab runner (deep halos, include/soil_slab.h) with spent walkers retired on every slab —
    a walker's first step, taken in the spawn, may carry it off its slab's rows or off the grid — against the
    single domain that walks everybody to the end: a 1024 x 512 grid in three slabs, two steps."""
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    from test_gpu_parallel import _run_world
    from test_gpu_parity import _close_but_for_stray_walks
    world, W, maxage, steps = 3, 512, 48, 2
    S = 1026 // world
    op = script_param(oracle.default_param())
    op.maxage = maxage
    pp = product_param(op)
    H = world * S
    retire("on")
    got = _run_world(world, S, W, pp, steps, maxage, pair=pair)
    retire("off")
    m = ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), pp, H * W // 8, seed=0)
    npar = soil.noise_t()
    npar.seed = 3.0
    npar.ext = [H, W]
    layers0 = np.zeros((H, W, 2), np.float32)
    layers0[..., 0] = to_np(soil.noise(silt.shape(H, W), npar, host=silt.gpu))
    m.set_layers(to_gpu(layers0))
    silt.set(m.rainfall, 1.0)
    for _ in range(steps):
        m.step()
    for k in got:
        want = to_np(getattr(m, k))
        _close_but_for_stray_walks(got[k], want, 1e-4, 1e-5 * (np.nanmax(np.abs(want)) + 1e-30), 1e-3, "slabs, " + k)


def test_first_steps_that_leave_the_grid(hip, oracle, retire):
    """A ramp that sends every walker of the rim off the grid with its first step: the spawn's own step ends those
    walks (erosion.cu:306), whatever the mode — same steps as the oracle with the retirement off, same planes on."""
    from soillib_amd import soil
    H, W = 96, 160
    N = H * W
    op = script_param(oracle.default_param())
    op.maxage = 32
    pp = product_param(op)
    scale = (20.0 / H, 20.0 / W, 400.0)
    x = np.arange(H, dtype=np.float32)[:, None] + np.zeros((1, W), np.float32)
    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0] = -0.01 * x - 0.003 * np.arange(W, dtype=np.float32)[None, :]      # downhill towards the far corner
    vel0 = np.zeros((H, W, 2), np.float32)
    mf, vf = np.zeros((H, W), np.float32), np.zeros((H, W, 2), np.float32)
    want = oracle.particles_debris(mf, vf, None, oracle.rng_seed(N, 6, 0), layers, vel0, None, scale, op)
    assert hip.soil_set_particle_mode(3) == 0
    retire("off")
    steps_off, mf_off, vf_off = _debris_launch(soil, layers, vel0, N, scale, pp)
    assert steps_off == want
    retire("on")
    steps_on, mf_on, vf_on = _debris_launch(soil, layers, vel0, N, scale, pp)
    assert steps_on <= want
    for got, ref, what in ((mf_on, mf, "mass flux"), (vf_on, vf, "velocity flux"), (mf_off, mf, "mass flux, off"),
                           (vf_off, vf, "velocity flux, off")):
        _same_planes(got, ref, what, rtol=1e-4)
