"""The Python surface a user of the reference finds after `import silt` / `import soillib as soil`:
which names exist, what the legacy names mean, and that every legacy entry point does the work
of its live counterpart — each on a small set-up of this file's own and checked against the oracle.

What is covered, by reference binding (python/source/model.cpp) or by the legacy scripts'
vocabulary (the live module no longer binds those names, their meaning is
reconstructed in soillib_amd/legacy.py):

  names        the live module's functions and types (model.cpp:23-60, 148-227, 237-421), the legacy
               containers and helpers, the silt runtime names
  param_t      legacy attribute -> live field, one pair at a time, both directions
  erode        map_t / data_t / param_t(samples) / erode: one whole step against the oracle's
               composition of the same step, track planes zero afterwards, steps numbered on
  buffer/index data_t(elem), buffer[:] = v, index(res).elem(), set / multiply / clamp
  resize       a scale-up of every plane of a model followed by a step on the finer grid
  graphs       direction / random_weighted / accumulate / accumulate_decay / multiflow
  normal       CPU tensors (BASELINE config 1's placement), timer
"""
import numpy as np
import pytest

from util import assert_bit_equal, assert_receivers_close, product_param, to_np

# --------------------------------------------------------------------------------- names (no GPU)

LIVE_FUNCTIONS = (
    "direction steepest random_weighted slope accumulate accumulate_decay gradient negslope laplacian "
    "gaussian_blur solve_uniform transport_fluvial transport_debris mass_transfer mass_creep layer_merge "
    "albedo_stratum albedo_layer albedo_discharge noise normal").split()
LIVE_TYPES = "param_t noise_t edge timer".split()
LEGACY_NAMES = "map_t data_t erode multiply clamp resize index buffer set float32 float64 int32 cpu gpu shape tensor".split()
IO_NAMES = "tiff geotiff geotiff_meta mesh util".split()
SILT_NAMES = "shape tensor float32 float64 int32 rng cpu gpu set add multiply clone seed".split()

# legacy attribute of example-era param_t -> field of soil::param_t (erosion.hpp:20-56); SURVEY.md 8a
LEGACY_PARAM = [
    ("viscosity", "viscosityWater"), ("bedShear", "bedShearWater"),
    ("suspensionRate", "suspensionRateFluvial"), ("depositionRate", "depositionRateFluvial"),
    ("critSlope", "critSlopeBedrock"), ("debrisCreepRate", "landslideRateDebris"),
    ("debrisSuspensionRate", "suspensionRateDebris"), ("debrisDepositionRate", "depositionRateDebris"),
    ("debrisYieldStress", "yieldStress"), ("debrisDensity", "densityDebris"),
    ("debrisViscosity", "viscosityDebris"), ("debrisBedShear", "bedShearDebris"),
]
SHARED_PARAM = "timeStep maxage lrate gravity uplift rainfall evapRate fluvialExponent exitSlope".split()


def test_module_names():
    import silt
    import soillib as soil
    for name in LIVE_FUNCTIONS + LEGACY_NAMES[2:9]:
        assert callable(getattr(soil, name)), name
    for name in LIVE_TYPES + LEGACY_NAMES[:2] + IO_NAMES + LEGACY_NAMES[9:]:
        assert hasattr(soil, name), name
    for name in SILT_NAMES:
        assert hasattr(silt, name), name
    assert (soil.d4, soil.d8) == (soil.edge.d4, soil.edge.d8) and soil.d4 != soil.d8
    assert len({soil.s, soil.ms, soil.us, soil.ns}) == 4 and soil.float32 is silt.float32 and soil.gpu is silt.gpu


@pytest.mark.parametrize("legacy,live", LEGACY_PARAM)
def test_legacy_param_name_is_the_live_field(legacy, live):
    import soillib as soil
    from soillib_amd import soil as live_module
    p = soil.param_t()
    q = live_module.param_t()
    assert getattr(p, legacy) == getattr(q, live)          # same default through either name
    setattr(p, legacy, 0.375)
    assert getattr(p, live) == 0.375 and getattr(p, legacy) == 0.375
    setattr(p, live, 1.625)
    assert getattr(p, legacy) == 1.625
    for other_legacy, other_live in LEGACY_PARAM:          # and nothing else moved
        if other_live != live:
            assert getattr(p, other_legacy) == getattr(q, other_live), other_legacy


def test_param_fields_shared_by_both_vocabularies_and_samples():
    import soillib as soil
    p = soil.param_t()
    for k, name in enumerate(SHARED_PARAM):
        setattr(p, name, 3 + k)
        assert getattr(p, name) == 3 + k, name
    assert p.samples == 8192                               # the script-era default particle count
    p.samples = 1000.0
    assert p.samples == 1000 and isinstance(p.samples, int)
    with pytest.raises(AttributeError):
        p.no_such_field = 1.0
    with pytest.raises(AttributeError):
        p.no_such_field


# --------------------------------------------------------------------------------------- GPU

def _planes(soil, silt, dims, value=0.0):
    t = silt.tensor(silt.float32, silt.shape(*dims), silt.gpu)
    silt.set(t, value)
    return t


def _legacy_model(soil, silt, oracle, H, W, scale, sediment, rain):
    bed = oracle.noise(H, W, seed=5.0, ext=(0.75 * H, 0.75 * W))
    model = soil.map_t(silt.shape(H, W), scale)
    model.height = silt.tensor.from_numpy(bed).gpu()
    model.sediment = _planes(soil, silt, (H, W), sediment)
    model.rainfall = _planes(soil, silt, (H, W), rain)
    model.uplift = _planes(soil, silt, (H, W), 0.0)

    def fields():
        d = soil.data_t(silt.shape(H, W))
        d.discharge, d.mass, d.debris = (_planes(soil, silt, (H, W)) for _ in range(3))
        d.momentum, d.debris_momentum = (_planes(soil, silt, (H, W, 2)) for _ in range(2))
        return d
    return model, fields(), fields(), bed


@pytest.mark.gpu
def test_erode_is_one_step_of_the_live_operators(hip, oracle):
    """soil.erode(model, data, track, param[, steps]) on the legacy containers = seed the streams,
    transport_fluvial, transport_debris, the cell phase — here against the oracle's composition of
    the same step (the legacy names select the same parameters the live ones do)."""
    import silt
    import soillib as soil
    H, W, scale = 96, 160, (0.2, 0.125, 3.0)
    model, data, track, bed = _legacy_model(soil, silt, oracle, H, W, scale, sediment=0.02, rain=0.7)
    N = H * W // 8
    p = soil.param_t()
    p.samples, p.maxage, p.timeStep = N, 40, 500.0
    p.critSlope, p.suspensionRate, p.debrisYieldStress = 0.06, 0.0006, 0.002
    op = oracle.default_param()
    op.maxage, op.timeStep = 40, 500.0
    op.critSlopeBedrock, op.suspensionRateFluvial, op.yieldStress = 0.06, 0.0006, 0.002

    soil.particle_steps(reset=True)
    soil.erode(model, data, track, p)                      # steps defaults to 1
    walked = soil.particle_steps(reset=True)

    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0], layers[..., 1] = bed, 0.02
    rain = np.full((H, W), 0.7, np.float32)
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    rng = oracle.rng_seed(N, 0, 0)
    wf, mf, vf, df, dvf = z1(), z1(), z2(), z1(), z2()
    steps = oracle.particles_fluvial(wf, mf, vf, None, rng, layers, rain, z1(), z2(), None, scale, op)
    steps += oracle.particles_debris(df, dvf, None, rng, layers, z2(), None, scale, op)
    want = oracle.erode_cells(layers, z1(), rain, wf, mf, vf, df, dvf, scale, op)
    assert walked == steps > N
    tol = lambda a: 1e-5 * (np.nanmax(np.abs(a)) + 1e-30)
    np.testing.assert_allclose(to_np(model.height), want["layers_next"][..., 0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(to_np(model.sediment), want["layers_next"][..., 1], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(to_np(data.discharge), want["waterHeight"], rtol=1e-4, atol=tol(want["waterHeight"]))
    np.testing.assert_allclose(to_np(data.mass), want["mass"], rtol=1e-4, atol=tol(want["mass"]))
    np.testing.assert_allclose(to_np(data.momentum), want["velocity"], rtol=1e-4, atol=tol(want["velocity"]))
    np.testing.assert_allclose(to_np(data.debris), want["debris"], rtol=1e-4, atol=tol(want["debris"]))
    assert np.abs(to_np(model.height) - bed).max() > 0
    for t in (track.discharge, track.mass, track.momentum, track.debris, track.debris_momentum):
        assert (to_np(t) == 0).all()                       # the flux planes come back zeroed
    # more steps: numbered on (fresh streams each), terrain stays finite; bad planes are refused
    soil.erode(model, data, track, p, 3)
    assert model._step_index == 4 and np.isfinite(to_np(model.height)).all()
    data.mass = _planes(soil, silt, (H, W // 2))
    with pytest.raises(ValueError):
        soil.erode(model, data, track, p)
    data.mass = silt.tensor(silt.float32, silt.shape(H, W), silt.cpu)
    with pytest.raises(ValueError):
        soil.erode(model, data, track, p)


@pytest.mark.gpu
def test_flat_buffers_index_and_scalar_helpers(hip):
    import soillib as soil
    idx = soil.index([24, 40])
    assert idx.elem() == 960 and (idx[0], idx[1]) == (24, 40)
    buf = soil.buffer(soil.float32, idx.elem(), soil.gpu)
    buf[:] = 2.5
    assert buf.host is soil.gpu and buf.elem() == 960 and (to_np(buf) == 2.5).all()
    assert soil.multiply(buf, -2.0) is buf and (to_np(buf) == -5.0).all()
    soil.set(buf, 0.125)
    assert (to_np(buf) == 0.125).all()
    host = soil.buffer(soil.float32, 7)                    # default placement: the host
    host[:] = [-3, -1, 0, 0.25, 0.5, 2, 9]
    assert host.host is soil.cpu and soil.clamp(host, -1.0, 0.5) is host
    assert host.numpy().tolist() == [-1, -1, 0, 0.25, 0.5, 0.5, 0.5]
    with pytest.raises(Exception):
        soil.clamp(buf, 0.0, 1.0)                          # device tensors: not part of the legacy helper
    d = soil.data_t(idx.elem())                            # data_t(elem): allocated, zeroed, on the device
    for name, n in (("discharge", 960), ("mass", 960), ("debris", 960), ("momentum", 1920), ("debris_momentum", 1920)):
        t = getattr(d, name)
        assert t.host is soil.gpu and t.elem() == n and (to_np(t) == 0).all(), name
    d.momentum[:] = [1.5, -2.0]                            # a vec2 fill of an interleaved plane
    assert (to_np(d.momentum).reshape(-1, 2) == [1.5, -2.0]).all()
    unbound = soil.data_t(soil.shape(24, 40))              # data_t(shape): the caller binds the planes
    assert unbound.discharge is None and unbound.debris_momentum is None
    timer = soil.timer(soil.us)
    with timer:
        soil.multiply(buf, 1.0)
    assert timer.count >= 0


@pytest.mark.gpu
def test_resize_carries_a_model_to_a_finer_grid(hip, oracle):
    """soil.resize(dst, src, newres, oldres) plane by plane (1 and 2 channels) against the oracle's
    resampling, then a step of soil.erode on the resized model."""
    import soillib as soil
    old, new = (40, 56), (72, 100)
    coarse = oracle.noise(old[0], old[1], seed=2.0, ext=(30.0, 42.0))
    two = np.stack([coarse, -2.0 * coarse], axis=-1).astype(np.float32)
    src1 = soil.tensor.from_numpy(coarse.reshape(-1)).gpu()
    src2 = soil.tensor.from_numpy(two.reshape(-1)).gpu()
    idx = soil.index(new)
    dst1 = soil.buffer(soil.float32, idx.elem(), soil.gpu)
    dst2 = soil.buffer(soil.float32, 2 * idx.elem(), soil.gpu)
    assert soil.resize(dst1, src1, new, old) is dst1
    soil.resize(dst2, src2, new, old)
    assert_bit_equal(to_np(dst1).reshape(new), oracle.resize(coarse, new), "resize, one channel")
    assert_bit_equal(to_np(dst2).reshape(new + (2,)), oracle.resize(two, new), "resize, two channels")
    with pytest.raises(ValueError):
        soil.resize(dst1, src2, new, old)                  # channel counts differ
    with pytest.raises(Exception):
        soil.resize(dst1, soil.tensor.from_numpy(coarse.reshape(-1)), new, old)   # a host tensor

    model = soil.map_t(idx, [20.0 / new[0], 20.0 / new[1], 4.0])
    model.height = dst1
    for name, v in (("sediment", 0.0), ("rainfall", 1.0), ("uplift", 0.0)):
        t = soil.buffer(soil.float32, idx.elem(), soil.gpu)
        t[:] = v
        setattr(model, name, t)
    data, track = soil.data_t(idx.elem()), soil.data_t(idx.elem())
    p = soil.param_t()
    p.samples, p.maxage, p.timeStep = idx.elem() // 8, 24, 100.0
    before = to_np(model.height).copy()
    soil.erode(model, data, track, p, 2)
    after = to_np(model.height)
    assert after.shape == (idx.elem(),) and np.isfinite(after).all() and np.abs(after - before).max() > 0
    assert np.nanmax(to_np(data.discharge)) > 0


@pytest.mark.gpu
def test_flow_graph_operators_and_the_multiflow_mean(hip, oracle):
    """random_weighted + accumulate averaged over realisations on the host (what a multiple-flow
    script does call by call) = soil.multiflow on the device, whole and in shards; every realisation's
    graph and sums against the oracle."""
    import silt
    import soillib as soil
    H, W, K, T = 88, 120, 6, 4.0
    dem = (oracle.noise(H, W, seed=7.0, ext=(60.0, 90.0)) * 80.0).astype(np.float32)
    height = silt.tensor.from_numpy(dem).gpu()
    rain_np = np.full((H, W), 1.5, np.float32)
    rain = silt.tensor.from_numpy(rain_np).gpu()
    mean = np.zeros((H, W))
    for k in range(K):
        graph = soil.random_weighted(height, soil.d8, 11, k, T)
        g = to_np(graph)
        assert_receivers_close(oracle, g, oracle.random_weighted(dem, 1, 11, k, T), dem, 8, 11, k, T)
        acc = to_np(soil.accumulate(graph, rain, soil.d8))
        assert_bit_equal(acc, oracle.accumulate(g, rain_np, 1), "accumulate, realisation %d" % k)
        assert acc[g < 0].sum() == pytest.approx(1.5 * H * W, rel=1e-6)      # the outlets drain everything
        mean += acc / float(K)
    np.testing.assert_allclose(to_np(soil.multiflow(height, rain, K, T, soil.d8, seed=11)), mean, rtol=1e-13)
    thirds = soil.multiflow(height, rain, K, T, soil.d8, seed=11, first=0, stride=3)
    for first in (1, 2):
        soil.multiflow(height, rain, K, T, soil.d8, seed=11, first=first, stride=3, out=thirds)
    np.testing.assert_allclose(to_np(thirds), mean, rtol=1e-13)

    codes = to_np(soil.direction(height, soil.d4))
    assert_bit_equal(codes, oracle.direction(dem, 0), "direction d4")
    assert codes.max() <= 3 and codes.min() == -1
    graph = soil.steepest(height, soil.d8)
    decay_np = (0.5 + 0.5 * np.random.default_rng(4).random((H, W))).astype(np.float32)
    got = to_np(soil.accumulate_decay(graph, rain, silt.tensor.from_numpy(decay_np).gpu(), soil.d8))
    want = oracle.accumulate(to_np(graph), rain_np, 1, decay=decay_np)
    assert_bit_equal(got, want, "accumulate_decay")        # powf_ on both sides
    with pytest.raises(Exception):
        soil.steepest(height.cpu(), soil.d8)               # a host tensor at a device operator: mismatch_host


@pytest.mark.gpu
def test_normal_of_a_host_tensor_and_the_relief_helper(hip, oracle):
    import silt
    import soillib as soil
    H, W = 70, 52
    dem = (oracle.noise(H, W, seed=9.0, ext=(35.0, 26.0)) * 30.0).astype(np.float32)
    scale = (2.0, 3.0, 0.5)
    host = soil.normal(silt.tensor.from_numpy(dem), scale)
    assert host.host is silt.cpu
    assert_bit_equal(host.numpy(), oracle.normal(dem, scale), "normal on the host")
    dev = soil.normal(silt.tensor.from_numpy(dem).gpu(), scale)
    assert dev.host is silt.gpu
    assert_bit_equal(to_np(dev), host.numpy(), "normal: device = host twin")
    n = host.numpy()
    np.testing.assert_allclose(np.linalg.norm(n, axis=-1), 1.0, rtol=1e-6)
    shade = soil.util.relief_shade(dem, n)
    assert shade.shape == (H, W) and np.isfinite(shade).all()
