"""The call sequences of the reference's acceptance scripts, run against
`import silt` / `import soillib as soil` exactly as those scripts spell them
(the scripts themselves live in the reference tree, which does not travel to the
GPU box, so their API usage is re-enacted here statement for statement).

  example/erosion_gpu.py    legacy map_t / data_t / param_t names / soil.erode
  example/dem_multiflow.py  random_weighted + accumulate loop, host-side averaging
  example/dem_process.py    direction / random_weighted / accumulate_decay
  example/tiff_normal.py    soil.normal on a CPU tensor
  example/erosion_gpu_multiscale.py   legacy index / buffer / data_t(elem) / resize / clamp
"""
import numpy as np
import pytest

from util import assert_receivers_close

pytestmark = pytest.mark.gpu


def test_erosion_gpu_script_sequence(hip):
    import silt
    import soillib as soil

    def noise(shape, scale):                      # erosion_gpu.py:9-15
        noise_param = soil.noise_t()
        noise_param.ext = np.array([shape[0], shape[1]]) * scale
        noise_param.seed = 3
        tensor = soil.noise(shape, noise_param)
        soil.multiply(tensor, 1.0)
        return tensor.gpu()

    def full(value, shape, dtype=silt.float32, host=silt.cpu):   # :17-20
        tensor = silt.tensor(dtype, shape, host)
        silt.set(tensor, value)
        return tensor

    simres = np.array([256, 256])
    shape = silt.shape(*simres)
    wscale = np.array([20.0, 20.0, 4.0])
    nscale = np.array([20.0, 20.0])
    pscale = [wscale[0] / simres[0], wscale[1] / simres[1], wscale[2]]

    model = soil.map_t(shape, pscale)             # :48-56
    model.height = noise(shape, nscale / wscale[0:2])
    model.sediment = full(0.0, shape, dtype=silt.float32, host=silt.gpu)
    model.rainfall = full(1.0, shape, dtype=silt.float32, host=silt.gpu)
    model.uplift = full(0.0, shape, dtype=silt.float32, host=silt.gpu)

    def make_data():                              # :59-71
        d = soil.data_t(shape)
        d.discharge = full(0.0, shape, dtype=silt.float32, host=silt.gpu)
        d.mass = full(0.0, shape, dtype=silt.float32, host=silt.gpu)
        d.debris = full(0.0, shape, dtype=silt.float32, host=silt.gpu)
        d.momentum = full(0.0, silt.shape(*simres, 2), dtype=silt.float32, host=silt.gpu)
        d.debris_momentum = full(0.0, silt.shape(*simres, 2), dtype=silt.float32, host=silt.gpu)
        return d
    data, track = make_data(), make_data()

    param = soil.param_t()                        # :75-100 (legacy attribute names)
    param.timeStep = 1000.0
    param.samples = 8192
    param.maxage = 256
    param.lrate = 1
    param.gravity = 9.81
    param.uplift = 0.01
    param.rainfall = 1.0
    param.evapRate = 0.0005
    param.viscosity = 0.000001
    param.bedShear = 12.5
    param.suspensionRate = 0.0008
    param.depositionRate = 0.00001
    param.fluvialExponent = 0.01
    param.exitSlope = 0.025
    param.critSlope = 0.57
    param.debrisCreepRate = 0.0025
    param.debrisSuspensionRate = 0.00025
    param.debrisDepositionRate = 0.0001
    param.debrisYieldStress = 2E6
    param.debrisDensity = 2500.0
    param.debrisViscosity = 0.004
    param.debrisBedShear = 60 / 2500.0
    assert param.viscosityWater == pytest.approx(1e-6) and param.bedShearDebris == pytest.approx(0.024)

    h0 = model.height.cpu().numpy().copy()
    timer = soil.timer()                          # :102-106
    for i in range(8):
        with timer:
            soil.erode(model, data, track, param, 1)
        assert timer.count >= 0
    h1 = model.height.cpu().numpy()
    sed = model.sediment.cpu().numpy()
    dis = data.discharge.cpu().numpy()
    assert h1.shape == (256, 256) and np.isfinite(h1).all() and np.isfinite(sed).all()
    assert np.abs(h1 - h0).max() > 0              # the terrain eroded
    assert (sed >= -1e-6).all()                    # (-y*sz)/sz may miss -y by an ulp (erosion.cu:538-539)
    assert np.nanmax(dis) > 0 and np.isfinite(dis.ravel()[1:]).all()
    for t in (track.discharge, track.mass, track.momentum, track.debris, track.debris_momentum):
        assert (t.cpu().numpy() == 0).all()       # flux planes are left zeroed for the next step
    relief = soil.util.relief_shade(h1, soil.normal(model.height.cpu(), [1, 1, 1]).numpy())
    assert relief.shape == (256, 256) and np.isfinite(relief).all()


def test_dem_multiflow_script_sequence(hip, oracle):
    import silt
    import soillib as soil
    H = W = 128
    dem = oracle.noise(H, W, seed=1.0, ext=(float(H), float(W))) * 100.0
    tensor = silt.tensor.from_numpy(dem.astype(np.float32)).gpu()     # dem_multiflow.py:24-26
    shape = tensor.shape
    res = (shape[0], shape[1])
    rain = np.full(res, 1.0)
    rain = silt.tensor.from_numpy(rain.astype(np.float32)).gpu()      # :29-31
    multiflow = np.full(res, 0.0)
    t = soil.timer(soil.us)
    K, T = 16, 10.0
    with t:
        for k in range(K):                                            # :43-49
            flow = soil.random_weighted(tensor, soil.d8, 0, k, T)
            accumulation = soil.accumulate(flow, rain, soil.d8)
            multiflow += accumulation.cpu().numpy() / float(K)
    assert t.count > 0
    assert multiflow.min() >= 1.0 - 1e-6           # every cell holds at least its own rain
    # the same loop kept on the GPU (soil_hip.h: soil_multiflow), whole and in two shards
    mean = soil.multiflow(tensor, rain, K, T, soil.d8, seed=0)
    np.testing.assert_allclose(mean.cpu().numpy(), multiflow, rtol=1e-13, atol=0)
    halves = soil.multiflow(tensor, rain, K, T, soil.d8, seed=0, first=0, stride=2)
    soil.multiflow(tensor, rain, K, T, soil.d8, seed=0, first=1, stride=2, out=halves)
    np.testing.assert_allclose(halves.cpu().numpy(), multiflow, rtol=1e-13, atol=0)
    # total drained area is conserved in every realisation: outlets sum to H*W
    flow0 = soil.random_weighted(tensor, soil.d8, 0, 0, T)
    acc0 = soil.accumulate(flow0, rain, soil.d8).cpu().numpy()
    assert acc0[flow0.cpu().numpy() < 0].sum() == H * W
    # bit-exact against the oracle for one realisation
    assert_receivers_close(oracle, flow0.cpu().numpy(), oracle.random_weighted(dem, 1, 0, 0, T), dem, 8, 0, 0, T)


def test_dem_process_and_tiff_normal_sequences(hip, oracle):
    import silt
    import soillib as soil
    H = W = 96
    dem = oracle.noise(H, W, seed=2.0, ext=(float(H), float(W))) * 50.0
    tensor = silt.tensor.from_numpy(dem).gpu()
    res = (H, W)
    rain = silt.tensor.from_numpy(np.full(res, 1.0).astype(np.float32)).gpu()
    dirn = soil.direction(tensor, soil.d8)                            # dem_process.py:31
    flow = soil.random_weighted(tensor, soil.d8, 0, 0, 10.0)          # :33
    decay = silt.tensor.from_numpy(np.full(res, 0.9).astype(np.float32)).gpu()   # :35-36
    discharge = soil.accumulate_decay(flow, rain, decay, soil.d8)     # :38
    d = discharge.cpu().numpy()
    assert d.min() >= 1.0 - 1e-6 and np.isfinite(d).all()
    assert dirn.cpu().numpy().max() <= 7
    np.testing.assert_array_equal(
        d, oracle.accumulate(flow.cpu().numpy(), np.ones(res, np.float32), 1, decay=np.full(res, 0.9, np.float32)))
    assert_receivers_close(oracle, flow.cpu().numpy(), oracle.random_weighted(dem, 1, 0, 0, 10.0), dem, 8, 0, 0, 10.0)
    # tiff_normal.py:14 — normal map of a CPU tensor, remapped for display
    normal = soil.normal(tensor.cpu(), [1.0, 1.0, 1.0]).numpy()
    normal = 0.5 + 0.5 * normal
    assert normal.shape == (H, W, 3) and normal.min() >= 0 and normal.max() <= 1


def test_erosion_gpu_multiscale_script_sequence(hip, oracle):
    import soillib as soil
    simres = np.array([32, 32])                           # erosion_gpu_multiscale.py:28-34
    wscale = np.array([20.0, 20.0, 4.0])
    nscale = np.array([20.0, 20.0])
    pscale = [wscale[0] / simres[0], wscale[1] / simres[1], wscale[2]]
    noise_param = soil.noise_t()                          # :36-38
    noise_param.ext = simres * nscale / wscale[0:2]
    noise_param.seed = 3
    index = soil.index(simres)                            # :40-42
    height = soil.noise(index, noise_param)
    soil.multiply(height, 1.0)
    sediment = soil.buffer(soil.float32, index.elem(), soil.gpu)   # :44-45
    sediment[:] = 0.0
    model = soil.map_t(index, pscale)                     # :49-59
    model.height = height.gpu()
    model.sediment = sediment.gpu()
    model.rainfall = soil.buffer(soil.float32, index.elem(), soil.gpu)
    soil.set(model.rainfall, 1.0)
    uplift = soil.noise(index, noise_param)
    soil.clamp(uplift, 0.0, 1.0)
    assert uplift.numpy().min() >= 0.0 and uplift.numpy().max() <= 1.0
    model.uplift = uplift.gpu()
    data = soil.data_t(index.elem())                      # :60-67
    track = soil.data_t(index.elem())
    data.discharge[:] = 0.0
    data.momentum[:] = [0.0, 0.0]
    data.mass[:] = 0.0
    data.debris[:] = 0.0
    data.debris_momentum[:] = [0.0, 0.0]
    param = soil.param_t()                                # :71-97 (abridged: same names)
    param.timeStep = 10.0
    param.samples = 2048
    param.maxage = 64
    param.uplift = 0.01
    param.suspensionRate = 0.0000008
    param.critSlope = 0.57
    timer = soil.timer()

    def scaleup(model, data, track, oldres, simres):      # :102-141
        index = soil.index(simres)
        pscale = [wscale[0] / simres[0], wscale[1] / simres[1], wscale[2]]
        planes = {}
        for name in ("height", "sediment", "rainfall", "uplift"):
            planes[name] = soil.buffer(soil.float32, index.elem(), soil.gpu)
            soil.resize(planes[name], getattr(model, name), simres, oldres)
        model = soil.map_t(index, pscale)
        for name, t in planes.items():
            setattr(model, name, t)
        newdata = soil.data_t(index.elem())
        newtrack = soil.data_t(index.elem())
        for name in ("mass", "discharge", "momentum", "debris_momentum", "debris"):
            soil.resize(getattr(newdata, name), getattr(data, name), simres, oldres)
        return model, newtrack, newdata, index, simres, pscale

    h_before = None
    for nextres, steps in (([32, 32], 3), ([64, 64], 2), ([100, 100], 2)):   # :143-160
        old_h = model.height.cpu().numpy().reshape(simres[0], simres[1])
        model, data, track, index, simres, pscale = scaleup(model, data, track, simres, nextres)
        up = model.height.cpu().numpy().reshape(nextres)
        np.testing.assert_array_equal(up, oracle.resize(old_h, tuple(nextres)))
        for _ in range(steps):
            with timer:
                soil.erode(model, data, track, param, 1)
        h_after = model.height.cpu().numpy()
        assert h_after.size == nextres[0] * nextres[1] and np.isfinite(h_after).all()
        assert np.abs(h_after.reshape(nextres) - up).max() > 0          # it eroded at this scale
        h_before = h_after
    assert h_before is not None and timer.count >= 0
