"""The particle step in FAST arithmetic (soil_set_particle_arith(1); csrc/erosion_particles_tiled.hip:
step_geom_fast / step_apply_fast): every quotient of the reference's step (erosion.cu:116-137,
:321-347) as numerator x v_rcp_f32(denominator), the norm as v_sqrt_f32, debris' mass attenuation on
v_exp_f32 — where the exact mode (the default, and the mode of every other parity test) reproduces the
IEEE quotients bit for bit.

A walk is chaotic in the last bit of its speed, so the statement about this mode is the statistical one
SURVEY.md 8 a5 makes about the transport as a whole, with the bounds tests/test_fma_tolerance.py sets for
the other build-to-build difference this path has (nvcc's contracted multiply-adds): against the ORACLE
(exact arithmetic, serial) on the same seeded inputs

  * the sums of the positive flux planes agree to 2e-3,
  * the sets of visited cells differ in at most 0.5 % of the visited cells,
  * the particle-step counts agree to 0.5 %,

and whole steps of the model stay within the same bounds of the exact mode's on the device.  The bench
line names the mode it ran (bench.py: config.particle_arithmetic).
"""
import numpy as np
import pytest

from util import product_param, rng_to_gpu, script_param, terrain, to_gpu, to_np

pytestmark = pytest.mark.gpu

TOL = {"flux_sum_rel": 2e-3, "visited_frac": 5e-3, "steps_rel": 5e-3}


@pytest.fixture
def fast(hip):
    """Fast arithmetic on the tiled shape whatever N; exact and automatic again afterwards."""
    assert hip.soil_get_particle_arith() == 0, "the exact step is the default of the process"
    assert hip.soil_set_particle_arith(1) == 0 and hip.soil_get_particle_arith() == 1
    assert hip.soil_set_particle_mode(3) == 0
    yield hip
    hip.soil_set_particle_mode(0)
    assert hip.soil_set_particle_arith(0) == 0


def _statistics(got, want, positive, what):
    for k in positive:
        a, b = np.nansum(got[k], dtype=np.float64), np.nansum(want[k], dtype=np.float64)
        assert abs(a - b) <= TOL["flux_sum_rel"] * abs(b), "%s %s: plane sums %.6e vs %.6e" % (what, k, a, b)
    for k in got:
        va, vb = got[k] != 0, want[k] != 0
        either = int(np.sum(va | vb))
        differ = int(np.sum(va != vb))
        assert differ <= TOL["visited_frac"] * either, "%s %s: %d of %d visited cells differ" % (what, k, differ, either)


def test_mode_switch_and_its_refusals(hip):
    from soillib_amd import soil
    assert soil.particle_arith() == "exact"
    assert hip.soil_set_particle_arith(2) != 0 and hip.soil_set_particle_arith(-1) != 0
    assert soil.particle_arith() == "exact"                  # a refused value changes nothing
    try:
        assert soil.particle_arith("fast") == "fast" and hip.soil_get_particle_arith() == 1
        with pytest.raises(ValueError):
            soil.particle_arith("faster")
    finally:
        assert soil.particle_arith("exact") == "exact"


@pytest.mark.parametrize("H,W,maxage,shape,rest", [(192, 192, 128, None, False), (320, 256, 200, "3", False),
                                                   # from rest (the model's first step: zero velocity and water
                                                   # planes): speeds are axis-aligned wherever one component of
                                                   # the downhill gradient is clamped to zero — direction
                                                   # components of EXACTLY zero, the face-time quotient's edge
                                                   (256, 192, 160, None, True)])
def test_fluvial_launch_fast_against_the_oracle(fast, oracle, monkeypatch, H, W, maxage, shape, rest):
    from soillib_amd import soil
    if shape:
        monkeypatch.setenv("SOIL_TILED_SHAPE", shape)       # the LDS-filling tiles large grids get
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers = terrain(oracle, H, W, sediment=0.02, rng_seed=5)
    p = script_param(oracle.default_param())
    p.maxage = maxage
    N = H * W // 8
    r = np.random.default_rng(3)
    vel = ((r.random((H, W, 2)) - 0.5) * 0.1).astype(np.float32)
    wh = (r.random((H, W)) * 0.01).astype(np.float32)
    if rest:
        vel[:], wh[:] = 0.0, 0.0
    rain = np.ones((H, W), np.float32)
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    want = dict(wf=z1(), mf=z1(), vf=z2())
    steps = oracle.particles_fluvial(want["wf"], want["mf"], want["vf"], None, oracle.rng_seed(N, 0, 0), layers, rain,
                                     wh, vel, None, scale, p)
    g = dict(wf=to_gpu(z1()), mf=to_gpu(z1()), vf=to_gpu(z2()))
    soil.particle_steps(reset=True)
    soil.transport_fluvial(to_gpu(layers), to_gpu(rain), to_gpu(wh), g["wf"], to_gpu(z1()), g["mf"], to_gpu(vel),
                           g["vf"], None, None, None, rng_to_gpu(oracle.rng_seed(N, 0, 0)), scale, product_param(p))
    walked = soil.particle_steps(reset=True)
    assert abs(walked - steps) <= TOL["steps_rel"] * steps, (walked, steps)
    got = {k: to_np(v) for k, v in g.items()}
    _statistics(got, want, ("wf", "mf"), "fluvial launch, fast arithmetic")
    # and the arithmetic really is another one: equal walks would give equal step counts and planes
    assert walked != steps or any((got[k] != want[k]).any() for k in got)


def test_debris_launch_fast_against_the_oracle(fast, oracle):
    from soillib_amd import soil
    H, W = 200, 168
    scale = (20.0 / H, 20.0 / W, 4.0)
    layers = terrain(oracle, H, W, sediment=0.01)
    p = script_param(oracle.default_param())
    p.maxage = 128
    p.critSlopeBedrock = 0.05          # landslides happen on the synthetic terrain
    p.yieldStress = 0.001
    N = H * W // 8
    vel = (np.random.default_rng(22).standard_normal((H, W, 2)) * 0.5).astype(np.float32)
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    want = dict(mf=z1(), vf=z2())
    steps = oracle.particles_debris(want["mf"], want["vf"], None, oracle.rng_seed(N, 6, 0), layers, vel, None, scale, p)
    assert steps > N and want["mf"].max() > 0
    g = dict(mf=to_gpu(z1()), vf=to_gpu(z2()))
    soil.particle_steps(reset=True)
    soil.transport_debris(to_gpu(layers), to_gpu(vel), g["vf"], to_gpu(z1()), g["mf"], None, None, None,
                          rng_to_gpu(oracle.rng_seed(N, 6, 0)), scale, product_param(p))
    walked = soil.particle_steps(reset=True)
    # (with spent walkers retired — soil_set_debris_retire(1), not the suite's mode — fewer steps are walked)
    assert abs(walked - steps) <= TOL["steps_rel"] * steps or (soil.debris_retire() == 1 and walked < steps), (walked, steps)
    _statistics({k: to_np(v) for k, v in g.items()}, want, ("mf",), "debris launch, fast arithmetic")


def test_whole_steps_fast_against_exact_on_the_device(hip, oracle):
    """Three steps of the model at 1024^2 (the launch shapes of BASELINE config 2) in either mode from the
    same state: water and sediment budgets within the launch bounds, the terrains within the bound the
    free-running exact steps hold against the oracle (tests/test_gpu_oracle_fullsize.py: a few strays)."""
    from soillib_amd import silt, soil
    from soillib_amd.erosion import ErosionModel
    S = 1024
    param = script_param(soil.param_t())
    out = {}
    for mode in ("exact", "fast"):
        soil.particle_arith(mode)
        try:
            m = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), param, S * S // 8, seed=1)
            q = soil.noise_t()
            q.seed = 3.0
            q.ext = [S, S]
            bed = soil.noise(silt.shape(S, S), q, host=silt.gpu)
            layers0 = np.zeros((S, S, 2), np.float32)
            layers0[..., 0] = to_np(bed)
            m.set_layers(to_gpu(layers0))
            silt.set(m.rainfall, 1.0)
            soil.particle_steps(reset=True)
            for _ in range(3):
                m.step()
            out[mode] = dict(steps=soil.particle_steps(reset=True), h=to_np(m.height), water=to_np(m.waterHeight),
                             sed=to_np(m.layers)[..., 1], h0=layers0[..., 0])
        finally:
            soil.particle_arith("exact")
    e, f = out["exact"], out["fast"]
    assert abs(f["steps"] - e["steps"]) <= TOL["steps_rel"] * e["steps"]
    assert np.isfinite(f["h"]).all() and np.isfinite(f["sed"]).all()
    for k in ("water", "sed"):
        a, b = np.nansum(f[k], dtype=np.float64), np.nansum(e[k], dtype=np.float64)
        assert abs(a - b) <= TOL["flux_sum_rel"] * abs(b), (k, a, b)
    dh_e, dh_f = e["h"] - e["h0"], f["h"] - f["h0"]
    assert np.abs(dh_e).max() > 0
    # the same erosion pattern: the change fields correlate, their volumes agree
    vol_e, vol_f = np.abs(dh_e).sum(dtype=np.float64), np.abs(dh_f).sum(dtype=np.float64)
    assert abs(vol_f - vol_e) <= 5e-3 * vol_e, (vol_f, vol_e)
    far = np.abs(f["h"] - e["h"]) > 1e-4 * (np.abs(dh_e).max())
    assert far.mean() <= 1e-2, "fast and exact terrains apart in %.2f %% of the cells" % (100 * far.mean())
