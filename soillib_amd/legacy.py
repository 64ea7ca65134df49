"""Legacy `soillib` names that example/erosion_gpu.py still calls.

The reference snapshot no longer binds them (they sit in a comment block,
python/source/model.cpp:62-143, and `soil::erode` has no definition), so their
meaning is reconstructed from the script and from the commented code
(SURVEY.md F3, §3.1, §8a):

    soil.map_t(shape, scale)   .height (bedrock) .sediment .uplift .rainfall   (model.cpp:67-97)
    soil.data_t(shape)         .discharge .momentum .mass .debris .debris_momentum (:103-140)
    soil.param_t()             legacy attribute names mapped onto the live ones (§8a)
    soil.erode(model, data, track, param, steps)                                (:142)
    soil.multiply(tensor, scalar) / soil.clamp(tensor, lo, hi)
and, for example/erosion_gpu_multiscale.py:
    soil.index(res), soil.buffer(dtype, elem, host) (`buf[:] = v`), soil.data_t(elem),
    soil.resize(dst, src, newres, oldres)

`erode` hands the caller's tensors to the library's step driver (soil_erode): `data.*`
are the transported fields, `track.*` the flux accumulators.
"""

from . import _abi, silt
from . import soil as _live

# legacy attribute -> live param_t field (live names erosion.hpp:20-56)
_LEGACY_PARAM = {
    "viscosity": "viscosityWater",
    "bedShear": "bedShearWater",
    "suspensionRate": "suspensionRateFluvial",
    "depositionRate": "depositionRateFluvial",
    "critSlope": "critSlopeBedrock",
    "debrisCreepRate": "landslideRateDebris",
    "debrisSuspensionRate": "suspensionRateDebris",
    "debrisDepositionRate": "depositionRateDebris",
    "debrisYieldStress": "yieldStress",
    "debrisDensity": "densityDebris",
    "debrisViscosity": "viscosityDebris",
    "debrisBedShear": "bedShearDebris",
}


class param_t(_live.param_t):
    """Live param_t that also answers to the legacy names of example/erosion_gpu.py:75-100.
    `samples` (particle count, = rng.elem() in the live API) is kept on the object."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "samples", 8192)

    def __setattr__(self, name, value):
        if name == "samples":
            object.__setattr__(self, "samples", int(value))
        else:
            super().__setattr__(_LEGACY_PARAM.get(name, name), value)

    def __getattr__(self, name):
        return super().__getattr__(_LEGACY_PARAM.get(name, name))


class map_t:
    """model.cpp:67-97: terrain planes + pixel scale."""

    def __init__(self, shape, scale):
        self.shape = shape if isinstance(shape, silt.shape) else silt.shape(*shape)
        self.scale = [float(v) for v in scale]
        self.height = None     # bedrock
        self.sediment = None
        self.uplift = None
        self.rainfall = None
        self._step_index = 0   # erosion steps taken so far (numbers the particle streams)


class data_t:
    """model.cpp:103-140: transported quantities (or their flux accumulators).
    `data_t(shape)` leaves the planes to the caller (erosion_gpu.py:59-71);
    `data_t(elem)` allocates them on the GPU (erosion_gpu_multiscale.py:60-67)."""

    def __init__(self, shape):
        self.discharge = None
        self.momentum = None
        self.mass = None
        self.debris = None
        self.debris_momentum = None
        if isinstance(shape, int):
            self.shape = silt.shape(shape)
            for name, dims in (("discharge", (shape,)), ("mass", (shape,)), ("debris", (shape,)),
                               ("momentum", (shape, 2)), ("debris_momentum", (shape, 2))):
                t = silt.tensor(silt.float32, silt.shape(*dims), silt.gpu)
                silt.set(t, 0.0)
                setattr(self, name, t)
        else:
            self.shape = shape if isinstance(shape, silt.shape) else silt.shape(*shape)


def index(res):
    """soil.index(res): the grid extent — a silt.shape today."""
    return silt.shape(*[int(v) for v in res])


def buffer(dtype, elem, host=silt.cpu):
    """soil.buffer(dtype, elem, host): a flat tensor; `buf[:] = v` fills it."""
    return silt.tensor(dtype, silt.shape(int(elem)), host)


def resize(dst, src, newres, oldres):
    """soil.resize(dst, src, newres, oldres) (erosion_gpu_multiscale.py:112-138): bilinear
    resampling of a plane of 1..3 interleaved channels on the GPU (soil_hip.h: soil_resize)."""
    Hn, Wn = int(newres[0]), int(newres[1])
    Ho, Wo = int(oldres[0]), int(oldres[1])
    for t in (dst, src):
        if t.host is not silt.gpu or t.type is not silt.float32:
            raise _abi.SoilError("resize: float32 silt.gpu tensors only")
    D = src.elem() // (Ho * Wo)
    if D * Ho * Wo != src.elem() or D * Hn * Wn != dst.elem():
        raise ValueError("resize: tensor sizes do not match the resolutions")
    _abi.check(_abi.lib().soil_resize(dst.c_ptr, src.c_ptr, Hn, Wn, Ho, Wo, D, _abi.stream()))
    return dst


def erode(model, data, track, param, steps=1):
    """soil.erode(model, data, track, param[, steps]) (example/erosion_gpu.py:102-106): one or
    more erosion steps on the legacy containers, in place.  The step loop,
    the re-seeding of the particle streams and the layer double buffer live in the library
    (soil_erode, csrc/erosion_step.hip); the planes are handed over as they are bound at the
    time of the call — a script may rebind model.uplift or resize its planes between calls."""
    H, W = model.shape[0], model.shape[1]
    if model.sediment is None:
        model.sediment = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
        silt.set(model.sediment, 0.0)
    m = _abi.ErodeModel()
    planes = {
        "height": (model.height, 1), "sediment": (model.sediment, 1), "uplift": (model.uplift, 1),
        "rainfall": (model.rainfall, 1), "discharge": (data.discharge, 1), "mass": (data.mass, 1),
        "momentum": (data.momentum, 2), "debris": (data.debris, 1),
        "debris_momentum": (data.debris_momentum, 2), "discharge_track": (track.discharge, 1),
        "mass_track": (track.mass, 1), "momentum_track": (track.momentum, 2),
        "debris_track": (track.debris, 1), "debris_momentum_track": (track.debris_momentum, 2),
    }
    for key, (t, channels) in planes.items():
        if t is None or t.host is not silt.gpu or t.type is not silt.float32:
            raise ValueError("erode: %s must be a float32 silt.gpu tensor" % key)
        if t.elem() != H * W * channels:
            raise ValueError("erode: %s does not hold %d x %d cells" % (key, H, W))
        setattr(m, key, t.ptr)
    first = getattr(model, "_step_index", 0)
    _abi.check(_abi.lib().soil_erode(_abi.C.byref(m), H, W, int(param.samples), 0, first, int(steps),
                                     _abi.vec(model.scale, 3), param._ref(), _abi.stream()))
    model._step_index = first + int(steps)


def multiply(tensor, value):
    """soil.multiply(tensor, scalar) (example/erosion_gpu.py:14)."""
    silt.multiply(tensor, value)
    return tensor


def clamp(tensor, lo, hi):
    """soil.clamp (example/erosion_gpu_multiscale.py): host tensors only — the
    multiscale driver is outside the hot path."""
    import numpy as np
    if tensor.host is not silt.cpu:
        raise _abi.SoilError("clamp: only silt.cpu tensors are supported")
    np.clip(tensor.numpy(), lo, hi, out=tensor.numpy())
    return tensor
