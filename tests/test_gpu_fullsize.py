"""Size-independent properties at BASELINE.json's sizes (the oracle finishes these in
minutes, not seconds, so the checks are between independent GPU paths and against
invariants of the domain):

  * particle transport at 4096^2 (N = 2.1 M, maxage 256): the reference's launch shape
    (direct) and the tiled one walk exactly the same number of steps and deposit the
    same flux up to fp32 summation order;
  * 8192^2 cell phase: the fused kernel equals the chain of stand-alone reference ops
    bit for bit (each of which is pinned to the oracle at small sizes);
  * 4096^2 D8 accumulation: every cell drains to exactly one outlet, so the outlets'
    sums add up to H*W (integers < 2^24: exact in fp32);
  * depression filling at 2048^2: idempotent, never below the input, no hole left;
  * device noise at 8192^2 equals the host generator on sampled rows.
"""
import ctypes as C

import numpy as np
import pytest

from util import script_param

pytestmark = pytest.mark.gpu


def _model(S, seed=3.0, maxage=256, n_div=8):
    from soillib_amd import _abi, silt, soil
    from soillib_amd.erosion import ErosionModel
    from util import script_param
    param = script_param(soil.param_t())
    param.maxage = maxage
    m = ErosionModel(S, S, (20.0 / S, 20.0 / S, 4.0), param, S * S // n_div, seed=0)
    p = soil.noise_t()
    p.seed = seed
    p.ext = [S, S]
    bed = soil.noise(silt.shape(S, S), p, host=silt.gpu)
    zero = silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
    silt.set(zero, 0.0)
    _abi.check(_abi.lib().soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, zero.c_ptr, S * S,
                                                  _abi.stream()))
    silt.set(m.rainfall, 1.0)
    silt.set(m.uplift, 0.0)
    return m


def test_particle_shapes_agree_at_4096(hip):
    from soillib_amd import silt, soil
    S = 4096
    m = _model(S)
    m.step()                                   # a velocity / water-height field to walk on
    m.step()
    results = {}
    for mode, name in ((1, "direct"), (3, "tiled")):
        assert hip.soil_set_particle_mode(mode) == 0
        for t in (m.waterFlux, m.massFlux, m.velocityFlux, m.debrisFlux, m.debrisVelocityFlux):
            silt.set(t, 0.0)
        m.seed_step()
        soil.particle_steps(reset=True)
        m.particles_fluvial()
        steps_f = soil.particle_steps(reset=True)
        m.particles_debris()
        steps_d = soil.particle_steps(reset=True)
        results[name] = (steps_f, steps_d, m.waterFlux.cpu().numpy().copy(),
                         m.velocityFlux.cpu().numpy().copy(), m.debrisFlux.cpu().numpy().copy())
    hip.soil_set_particle_mode(0)
    a, b = results["direct"], results["tiled"]
    from util import debris_steps_match
    assert a[0] == b[0] and debris_steps_match(b[1], a[1]) and a[0] > 10 * m.N     # same walks, step for step
    for x, y, what in ((a[2], b[2], "water flux"), (a[3], b[3], "velocity flux"),
                       (a[4], b[4], "debris flux")):
        scale = np.nanmax(np.abs(x))
        np.testing.assert_allclose(y, x, rtol=2e-5, atol=2e-6 * scale, err_msg=what)
        if what == "water flux":    # positive deposits only: no cancellation to an exact zero
            assert ((x != 0) == (y != 0)).all(), what + ": different set of visited cells"


def test_fused_cells_equal_reference_chain_at_8192(hip):
    from soillib_amd import silt, soil
    S = 8192
    m = _model(S, maxage=64, n_div=64)         # a short particle phase: flux planes with content
    m.seed_step()
    m.particles_fluvial()
    m.particles_debris()
    keep = {k: silt.clone(getattr(m, k)) for k in
            ("waterFlux", "massFlux", "velocityFlux", "debrisFlux", "debrisVelocityFlux")}
    m.cells_fused()
    fused = {k: getattr(m, k).cpu().numpy().copy() for k in
             ("layers_next", "height", "waterHeight", "mass", "velocity", "debris",
              "debrisVelocity")}
    for k in keep:
        assert (getattr(m, k).cpu().numpy() == 0).all(), k + " is left zeroed"
    # the same phase through the stand-alone ops (model.cpp:273-351 order)
    z1 = lambda: silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
    z2 = lambda: silt.tensor(silt.float32, silt.shape(S, S, 2), silt.gpu)
    wh, mass, vel, deb, dvel = z1(), z1(), z2(), z1(), z2()
    for t in (wh, mass, vel, deb, dvel):
        silt.set(t, 0.0)
    from soillib_amd import _abi
    sc = _abi.vec(m.scale, 3)
    # N = 0 particles: soil_transport_* then is the normalisation of the given flux planes
    _abi.check(hip.soil_transport_fluvial(
        m.layers.c_ptr, m.rainfall.c_ptr, wh.c_ptr, keep["waterFlux"].c_ptr, mass.c_ptr,
        keep["massFlux"].c_ptr, vel.c_ptr, keep["velocityFlux"].c_ptr, None, None, None, None, 0,
        S, S, sc, m.param._ref(), None))
    _abi.check(hip.soil_transport_debris(
        m.layers.c_ptr, dvel.c_ptr, keep["debrisVelocityFlux"].c_ptr, deb.c_ptr,
        keep["debrisFlux"].c_ptr, None, None, None, None, 0, S, S, sc, m.param._ref(), None))
    delta = z2()
    silt.set(delta, 0.0)
    soil.mass_transfer(delta, m.layers, m.uplift, wh, mass, vel, deb, dvel, None, None, None, None,
                       m.scale, m.param)
    soil.mass_creep(delta, m.layers, m.scale, m.param)
    layers = silt.clone(m.layers)
    silt.add(layers, delta)
    height = z1()
    soil.layer_merge(height, layers)
    for got, want, what in ((layers, "layers_next", "layers"), (height, "height", "height"),
                            (wh, "waterHeight", "waterHeight"), (mass, "mass", "mass"),
                            (vel, "velocity", "velocity"), (deb, "debris", "debris"),
                            (dvel, "debrisVelocity", "debrisVelocity")):
        a, b = got.cpu().numpy(), fused[want]
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), "%s: %d cells differ" % (what, (~same).sum())


def test_accumulation_drains_every_cell_at_4096(hip):
    from soillib_amd import silt, soil
    S = 4096
    p = soil.noise_t()
    p.seed = 5.0
    p.ext = [S, S]
    h = soil.noise(silt.shape(S, S), p, host=silt.gpu)
    silt.multiply(h, 100.0)
    rain = silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
    silt.set(rain, 1.0)
    for flow in (soil.steepest(h, soil.d8), soil.random_weighted(h, soil.d8, 0, 7, 10.0)):
        acc = soil.accumulate(flow, rain, soil.d8).cpu().numpy()
        f = flow.cpu().numpy()
        assert acc.min() >= 1.0
        assert acc[f < 0].astype(np.float64).sum() == S * S
        # a receiver holds at least its donors' areas + itself
        recv = f[f >= 0]
        assert (acc.reshape(-1)[recv] > acc[f >= 0]).all()


def test_accumulate_without_decay_arrays_equals_decay_of_ones_at_4096(hip):
    """BASELINE config 3's call (no decay tensor: no decay arrays, 32-bit offsets) against the general
    kernel fed a decay tensor of ones, on non-integer sources: bit for bit (graph.cu:577-583)."""
    from soillib_amd import silt, soil
    S = 4096
    p = soil.noise_t()
    p.seed = 11.0
    p.ext = [S, S]
    h = soil.noise(silt.shape(S, S), p, host=silt.gpu)
    silt.multiply(h, 100.0)
    src = soil.noise(silt.shape(S, S), p, host=silt.gpu)   # values in (-1, 1): sums that round
    ones = silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
    silt.set(ones, 1.0)
    for edge in (soil.d4, soil.d8):
        flow = soil.random_weighted(h, edge, 0, 3, 10.0)
        a = soil.accumulate(flow, src, edge).cpu().numpy()
        b = soil.accumulate_decay(flow, src, ones, edge).cpu().numpy()
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_random_weighted_receivers_at_2048(hip, oracle):
    """BASELINE config 3's graph maker at size against the oracle's exact statement: the DEM of
    dem_multiflow.py (heights of ~100 m, T = 10, D8), 4.2 M cells, two draws.  The receivers are equal
    but for a counted few, each of them a draw on a CDF edge (util.assert_receivers_close).
    (Widths that are not a multiple of four take the one-cell-per-thread kernel — the same operations on
    the same values: tests/test_gpu_parity.py::test_flow_maps_bit_exact runs both.)"""
    from soillib_amd import silt, soil
    from util import assert_receivers_close
    S = 2048
    p = soil.noise_t()
    p.seed = 5.0
    p.ext = [S, S]
    h = soil.noise(silt.shape(S, S), p, host=silt.gpu)
    silt.multiply(h, 100.0)
    hn = h.cpu().numpy()
    oracle.set_threads(16)
    differ = 0
    for off in (0, 511):
        got = soil.random_weighted(h, soil.d8, 0, off, 10.0).cpu().numpy()
        want = oracle.random_weighted(hn, 1, 0, off, 10.0)
        assert_receivers_close(oracle, got, want, hn, 8, 0, off, 10.0)
        differ += int((got != want).sum())
    print("random_weighted at 2048^2: %d of %d receivers differ from the oracle's" % (differ, 2 * S * S))


def test_fill_depressions_properties_at_2048(hip):
    from soillib_amd import silt, soil
    S = 2048
    p = soil.noise_t()
    p.seed = 9.0
    p.ext = [S, S]
    h = soil.noise(silt.shape(S, S), p, host=silt.gpu)
    silt.multiply(h, 100.0)
    filled = soil.fill_depressions(h, soil.d8)
    a, b = filled.cpu().numpy(), h.cpu().numpy()
    assert (a >= b).all() and (a > b).any()
    again = soil.fill_depressions(filled, soil.d8).cpu().numpy()
    assert (again.view(np.uint32) == a.view(np.uint32)).all()
    # no interior cell lies strictly below all of its 8 neighbours
    c = a[1:-1, 1:-1]
    nb = np.minimum.reduce([a[:-2, 1:-1], a[2:, 1:-1], a[1:-1, :-2], a[1:-1, 2:],
                            a[:-2, :-2], a[:-2, 2:], a[2:, :-2], a[2:, 2:]])
    assert (nb <= c).all()


def test_device_noise_equals_host_generator_at_8192(hip):
    from soillib_amd import _abi, silt, soil
    S = 8192
    p = soil.noise_t()
    p.seed = 3.0
    p.ext = [S, S]
    dev = soil.noise(silt.shape(S, S), p, host=silt.gpu).cpu().numpy()
    rows = (0, 1, 4095, 8191)
    for x0 in rows:
        win = silt.tensor(silt.float32, silt.shape(1, S), silt.gpu)
        _abi.check(hip.soil_noise_window(win.c_ptr, 1, S, x0, C.byref(p._c), None))
        assert (win.cpu().numpy()[0].view(np.uint32) == dev[x0].view(np.uint32)).all()
    small = soil.noise_t()
    small.seed = 3.0
    small.ext = [S, S]
    host = soil.noise(silt.shape(64, 64), small).numpy()          # host twin, a corner of a 64^2 grid
    dev64 = soil.noise(silt.shape(64, 64), small, host=silt.gpu).cpu().numpy()
    assert (host.view(np.uint32) == dev64.view(np.uint32)).all()


def test_fill_depressions_properties_at_4096(hip):
    """BASELINE config 3 as written: pit fill of a 4096^2 DEM.  Size-independent properties of the
    priority-flood surface: on or above the DEM, idempotent bit for bit, no interior cell strictly
    below all of its 8 neighbours, untouched where the DEM already drains (every raised cell sits
    on a flat or rising path: it equals the minimum of its neighbours' fill or its own height)."""
    from soillib_amd import silt, soil
    S = 4096
    p = soil.noise_t()
    p.seed = 9.0
    p.ext = [S, S]
    h = soil.noise(silt.shape(S, S), p, host=silt.gpu)
    silt.multiply(h, 100.0)
    filled = soil.fill_depressions(h, soil.d8)
    a, b = filled.cpu().numpy(), h.cpu().numpy()
    assert (a >= b).all() and (a > b).any()
    again = soil.fill_depressions(filled, soil.d8).cpu().numpy()
    assert (again.view(np.uint32) == a.view(np.uint32)).all()
    c = a[1:-1, 1:-1]
    nb = np.minimum.reduce([a[:-2, 1:-1], a[2:, 1:-1], a[1:-1, :-2], a[1:-1, 2:],
                            a[:-2, :-2], a[:-2, 2:], a[2:, :-2], a[2:, 2:]])
    assert (nb <= c).all()
    # the fixed point itself: w = max(z, min over neighbours of w) in the interior, w = z on the rim
    assert (c.view(np.uint32) == np.maximum(b[1:-1, 1:-1], nb).view(np.uint32)).all()
    for rim in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
        assert (a[rim].view(np.uint32) == b[rim].view(np.uint32)).all()


def test_whole_step_at_8192_direct_equals_tiled(hip):
    """BASELINE config 4's grid through the product's tiled transport and through the reference's
    launch shape (one lane per streamline, global atomics): the same number of particle steps
    (identical trajectories), flux planes equal up to fp32 summation order, and a cell phase that
    leaves the flux planes zero and the terrain finite."""
    from soillib_amd import _abi, silt, soil
    from soillib_amd.erosion import ErosionModel
    S = 8192
    N = S * S // 8
    pp = script_param(soil.param_t())
    scale = (20.0 / S, 20.0 / S, 4.0)
    npar = soil.noise_t()
    npar.seed = 3.0
    npar.ext = [S, S]
    bed = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
    m = ErosionModel(S, S, scale, pp, N, seed=0)
    _abi.check(hip.soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, None, bed.elem(), None))
    silt.set(m.rainfall, 1.0)
    m.step()                       # a first step so that velocity / water fields are live
    names = ("waterFlux", "massFlux", "velocityFlux", "debrisFlux", "debrisVelocityFlux")
    got = {}
    try:
        for mode, label in ((3, "tiled"), (1, "direct")):
            _abi.check(hip.soil_set_particle_mode(mode))
            for n in names:
                silt.set(getattr(m, n), 0.0)
            soil.particle_steps(reset=True)
            m.seed_step()
            m.particles_fluvial()
            m.particles_debris()
            _abi.check(hip.soil_device_synchronize())
            got[label] = (soil.particle_steps(reset=True), {n: getattr(m, n).cpu().numpy() for n in names})
    finally:
        _abi.check(hip.soil_set_particle_mode(0))
    from util import debris_steps_match
    assert debris_steps_match(got["tiled"][0], got["direct"][0]) and got["direct"][0] > 2.0e9
    for n in names:
        a, b = got["tiled"][1][n], got["direct"][1][n]
        sc = np.nanmax(np.abs(b)) + 1e-30
        # a channel cell sums 1e4..1e5 deposits at this size: the tolerance of the oracle tests at size
        bad = ~(np.abs(a - b) <= 1e-4 * np.abs(b) + 2e-5 * sc) & ~(np.isnan(a) & np.isnan(b))
        assert bad.sum() == 0, "%s: %d cells differ" % (n, bad.sum())
        # conservation, whatever the order of the additions: plane sums in double
        sa, sb = np.nansum(a.astype(np.float64)), np.nansum(b.astype(np.float64))
        ref = np.nansum(np.abs(b).astype(np.float64)) + 1e-300
        assert abs(sa - sb) <= 1e-6 * ref, (n, sa, sb)
    m.cells_fused()
    _abi.check(hip.soil_device_synchronize())
    for n in names:
        assert not getattr(m, n).cpu().numpy().any(), n + " not re-zeroed"
    hh = m.height.cpu().numpy()
    assert np.isfinite(hh).sum() >= hh.size - 1     # cell (0,0) may hold the NaN walkers' deposit


def test_config1_geotiff_to_gpu_normal(hip, oracle, tmp_path):
    """BASELINE config 1 with the tensor moved to the device: a 256^2 GeoTIFF written and read by
    the library's own codec, .gpu(), soil.normal on the device == the CPU twin == the oracle."""
    import silt
    import soillib as soil
    H = W = 256
    p = soil.noise_t()
    p.seed = 3.0
    p.ext = [H, W]
    height = soil.noise(silt.shape(H, W), p)
    g = soil.geotiff(height)
    g.meta.scale = [2.0, 2.0, 80.0]
    path = str(tmp_path / "dem_256.tiff")
    g.write(path)
    image = soil.geotiff(path)
    assert (image.tensor.numpy().reshape(-1).view(np.uint32) == height.numpy().reshape(-1).view(np.uint32)).all()
    want = oracle.normal(height.numpy(), (2.0, 2.0, 80.0))
    cpu = soil.normal(image.tensor, image.meta.scale).numpy()
    dev = soil.normal(image.tensor.gpu(), image.meta.scale).cpu().numpy()
    np.testing.assert_array_equal(cpu, want)
    np.testing.assert_array_equal(dev, want)
