"""Host-side mirror of the reference's `soillib` Python module for the erosion path.

Same names, argument order and meaning as the nanobind module
(python/source/model.cpp:23-60 param_t, :148-151 edge, :157-203 graph/stencil
ops returning new tensors, :209-227 solve_uniform, :237-407 in-place erosion /
albedo ops returning None, :413-421 noise_t / noise; python/source/util.cpp:47-73
timer).  Every op forwards to the C ABI (include/soil_hip.h); tensors are
`silt` tensors (soillib_amd.silt).  GPU ops given a CPU tensor raise, like
silt::error::mismatch_host in the reference (graph.cu:75-76).
"""
import ctypes as C
import time

import numpy as np

from . import _abi, silt

# ---------------------------------------------------------------- edge enum


class edge:
    """soil::edge_t, graph.hpp:11-14 / model.cpp:148-151."""
    d4 = _abi.D4
    d8 = _abi.D8


d4, d8 = edge.d4, edge.d8  # export_values()


# ------------------------------------------------------------------ param_t

class param_t:
    """soil::param_t (erosion.hpp:17-58), bound field by field at model.cpp:23-60."""

    _FIELDS = ("maxage",) + _abi._PARAM_FLOATS

    def __init__(self):
        object.__setattr__(self, "_c", _abi.Param())
        _abi.lib().soil_param_default(C.byref(self._c))

    def __getattr__(self, name):
        if name == "force":
            return [self._c.force[0], self._c.force[1]]
        if name in param_t._FIELDS:
            return getattr(self._c, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name == "force":
            self._c.force[0], self._c.force[1] = float(value[0]), float(value[1])
        elif name == "maxage":
            self._c.maxage = int(value)
        elif name in param_t._FIELDS:
            setattr(self._c, name, float(value))
        else:
            raise AttributeError("param_t has no field %r" % name)

    def _ref(self):
        return C.byref(self._c)


# ------------------------------------------------------------------ helpers

def _gpu(t, dtype, what):
    if not isinstance(t, silt.tensor):
        raise TypeError("%s: expected a silt.tensor" % what)
    if t.host is not silt.gpu:
        raise _abi.SoilError("mismatch_host: %s must be a silt.gpu tensor" % what)
    if t.type is not dtype:
        raise TypeError("%s: expected %s, got %s" % (what, dtype.name, t.type.name))
    return t.c_ptr


def _f(t, what):
    return _gpu(t, silt.float32, what)


def _opt_f(t, what):
    return None if t is None else _gpu(t, silt.float32, what)


def _hw(t):
    s = t.shape
    return s[0], s[1]


def _call(name, *args):
    _abi.check(getattr(_abi.lib(), name)(*args))


# --------------------------------------------------------- graph / stencils

def direction(height, edge_):
    """model.cpp:157-159 -> soil::direction (graph.cu:246-264)."""
    H, W = _hw(height)
    out = silt.tensor(silt.int32, silt.shape(H, W), silt.gpu)
    _call("soil_direction", out.c_ptr, _f(height, "height"), H, W, edge_, _abi.stream())
    return out


def steepest(height, edge_):
    """model.cpp:169-171 -> soil::steepest (graph.cu:73-91)."""
    H, W = _hw(height)
    out = silt.tensor(silt.int32, silt.shape(H, W), silt.gpu)
    _call("soil_steepest", out.c_ptr, _f(height, "height"), H, W, edge_, _abi.stream())
    return out


def random_weighted(height, edge_, seed, offset, T):
    """model.cpp:173-175 -> soil::random_weighted (graph.cu:175-195)."""
    H, W = _hw(height)
    out = silt.tensor(silt.int32, silt.shape(H, W), silt.gpu)
    _call("soil_random_weighted", out.c_ptr, _f(height, "height"), H, W, edge_, int(seed),
          int(offset), float(T), _abi.stream())
    return out


def slope(tensor, flow, scale):
    """model.cpp:161-163 -> soil::slope (graph.cu:297-311)."""
    H, W = _hw(tensor)
    out = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
    _call("soil_slope", out.c_ptr, _f(tensor, "tensor"), _gpu(flow, silt.int32, "flow"), H, W,
          _abi.vec(scale, 2), _abi.stream())
    return out


def accumulate(graph, field, edge_):
    """model.cpp:181-183 -> soil::accumulate (graph.cu:578-584)."""
    H, W = _hw(graph)
    out = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
    _call("soil_accumulate", out.c_ptr, _gpu(graph, silt.int32, "graph"), _f(field, "field"), None,
          H, W, edge_, _abi.stream())
    return out


def accumulate_decay(graph, field, decay, edge_):
    """model.cpp:185-187 -> soil::accumulate_decay (graph.cu:586-593)."""
    H, W = _hw(graph)
    out = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
    _call("soil_accumulate", out.c_ptr, _gpu(graph, silt.int32, "graph"), _f(field, "field"),
          _f(decay, "decay"), H, W, edge_, _abi.stream())
    return out


def fill_depressions(height, edge_=None):
    """Priority-flood surface of a DEM (soil_hip.h: soil_fill_depressions) — the
    conditioning step the reference leaves to pysheds (example/dem_condition.py:35-41)."""
    H, W = _hw(height)
    out = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
    _call("soil_fill_depressions", out.c_ptr, _f(height, "height"), H, W,
          int(d8 if edge_ is None else edge_), _abi.stream())
    return out


def multiflow(height, source, K, T, edge_=None, seed=0, first=0, stride=1, out=None):
    """Mean of `accumulate(random_weighted(height, edge, seed, k, T), source)` over the
    realisations k = first, first+stride, ... < K, each term divided by K in float32 and
    summed in float64 — the loop of example/dem_multiflow.py:43-49 kept on the GPU
    (soil_hip.h: soil_multiflow).  Returns (or adds into `out`) a float64 GPU tensor."""
    H, W = _hw(height)
    if out is None:
        out = silt.tensor(silt.float64, silt.shape(H, W), silt.gpu)
        _call("soil_set_f32", out.c_ptr, 0.0, 2 * H * W, _abi.stream())   # all-zero bits = 0.0
    e = d8 if edge_ is None else edge_
    _call("soil_multiflow", out.c_ptr, _f(height, "height"), _f(source, "source"), H, W, int(e),
          int(seed), int(first), int(stride), int(K), int(K), float(T), _abi.stream())
    return out


def gaussian_blur(tensor, sigma):
    """model.cpp:189-191 -> soil::gaussian_blur (filter.cu:72-91): blurs IN PLACE
    and returns its input handle (filter.cu:90)."""
    H, W = _hw(tensor)
    Cn = tensor.shape[2]
    scratch = silt.tensor(silt.float32, tensor.shape, silt.gpu)
    _call("soil_gaussian_blur", _f(tensor, "tensor"), scratch.c_ptr, H, W, Cn, float(sigma),
          _abi.stream())
    return tensor


def gradient(tensor, scale):
    """model.cpp:193-195 -> soil::gradient (grad.cu:89-97); returns (H, W, 2)."""
    H, W = _hw(tensor)
    out = silt.tensor(silt.float32, silt.shape(H, W, 2), silt.gpu)
    _call("soil_gradient", out.c_ptr, _f(tensor, "tensor"), H, W, _abi.vec(scale, 2),
          _abi.stream())
    return out


def laplacian(tensor, scale):
    """model.cpp:197-199 -> soil::laplacian (grad.cu:186-206); same shape as the input."""
    H, W = _hw(tensor)
    D = tensor.shape[2]
    out = silt.tensor(silt.float32, tensor.shape, silt.gpu)
    _call("soil_laplacian", out.c_ptr, _f(tensor, "tensor"), H, W, D, _abi.vec(scale, 2),
          _abi.stream())
    return out


def negslope(tensor, scale):
    """model.cpp:201-203 -> soil::negslope (grad.cu:133-141)."""
    H, W = _hw(tensor)
    out = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu)
    _call("soil_negslope", out.c_ptr, _f(tensor, "tensor"), H, W, _abi.vec(scale, 2),
          _abi.stream())
    return out


def normal(tensor, scale=(1.0, 1.0, 1.0)):
    """soil::op::normal (op/normal.hpp:19-39; example/tiff_normal.py:14).  CPU
    tensors are processed on the host like in the reference, GPU tensors in HBM."""
    s = tensor.shape
    if s.dim() != 2:
        raise ValueError("normal map can not be computed for non 2D-indexed buffers")  # :22-23
    H, W = s[0], s[1]
    if tensor.host is silt.gpu:
        out = silt.tensor(silt.float32, silt.shape(H, W, 3), silt.gpu)
        _call("soil_normal", out.c_ptr, _f(tensor, "tensor"), H, W, _abi.vec(scale, 3),
              _abi.stream())
        return out
    src = np.ascontiguousarray(tensor.numpy(), dtype=np.float32)
    dst = np.empty((H, W, 3), np.float32)
    _call("soil_normal_host", dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), H,
          W, _abi.vec(scale, 3))
    return silt.tensor._wrap_numpy(dst)


# ----------------------------------------------------------- solve_uniform

def solve_uniform(flow, source, decay, rng, scale, count):
    """model.cpp:209-227 -> soil::solve_uniform (path.cu:180-219)."""
    s = source.shape
    H, W, K = s[0], s[1], s[2]
    flux = silt.tensor(silt.float32, silt.shape(H, W, K), silt.gpu)
    _call("soil_solve_uniform", flux.c_ptr, _f(flow, "flow"), _f(source, "source"),
          _f(decay, "decay"), _gpu(rng, silt.rng, "rng"), rng.elem(), H, W, K, _abi.vec(scale, 2),
          int(count), _abi.stream())
    return flux


# ------------------------------------------------------------- erosion ops

def transport_fluvial(layers, rainfall, discharge, discharge_track, mass, mass_track, momentum,
                      momentum_track, albedo_bedrock, albedo_transport, albedo_surface, rng, scale,
                      param):
    """model.cpp:237-268 -> soil::transport_fluvial (erosion.cu:189-239)."""
    H, W = _hw(layers)
    _call("soil_transport_fluvial", _f(layers, "layers"), _f(rainfall, "rainfall"),
          _f(discharge, "discharge"), _f(discharge_track, "discharge_track"), _f(mass, "mass"),
          _f(mass_track, "mass_track"), _f(momentum, "momentum"),
          _f(momentum_track, "momentum_track"), _opt_f(albedo_bedrock, "albedo_bedrock"),
          _opt_f(albedo_transport, "albedo_transport"), _opt_f(albedo_surface, "albedo_surface"),
          _gpu(rng, silt.rng, "rng"), rng.elem(), H, W, _abi.vec(scale, 3), param._ref(),
          _abi.stream())


def transport_debris(layers, velocity, velocity_track, mass, mass_track, albedo_bedrock,
                     albedo_transport, albedo_surface, rng, scale, param):
    """model.cpp:270-295 -> soil::transport_debris (erosion.cu:395-436)."""
    H, W = _hw(layers)
    _call("soil_transport_debris", _f(layers, "layers"), _f(velocity, "velocity"),
          _f(velocity_track, "velocity_track"), _f(mass, "mass"), _f(mass_track, "mass_track"),
          _opt_f(albedo_bedrock, "albedo_bedrock"), _opt_f(albedo_transport, "albedo_transport"),
          _opt_f(albedo_surface, "albedo_surface"), _gpu(rng, silt.rng, "rng"), rng.elem(), H, W,
          _abi.vec(scale, 3), param._ref(), _abi.stream())


def mass_transfer(deltas, layers, uplift, discharge, mass, momentumFluvial, debris, momentumDebris,
                  albedo_bedrock, albedo_transport_fluvial, albedo_transport_debris,
                  albedo_surface, scale, param):
    """model.cpp:297-328 -> soil::mass_transfer (erosion.cu:576-611)."""
    H, W = _hw(uplift)
    _call("soil_mass_transfer", _f(deltas, "deltas"), _f(layers, "layers"), _f(uplift, "uplift"),
          _f(discharge, "discharge"), _f(mass, "mass"), _f(momentumFluvial, "momentumFluvial"),
          _f(debris, "debris"), _f(momentumDebris, "momentumDebris"),
          _opt_f(albedo_bedrock, "albedo_bedrock"),
          _opt_f(albedo_transport_fluvial, "albedo_transport_fluvial"),
          _opt_f(albedo_transport_debris, "albedo_transport_debris"),
          _opt_f(albedo_surface, "albedo_surface"), H, W, _abi.vec(scale, 3), param._ref(),
          _abi.stream())


def mass_creep(delta, layers, scale, param):
    """model.cpp:330-341 -> soil::mass_creep (erosion.cu:712-727)."""
    H, W = _hw(layers)
    _call("soil_mass_creep", _f(delta, "delta"), _f(layers, "layers"), H, W, _abi.vec(scale, 3),
          param._ref(), _abi.stream())


def particle_steps(reset=True):
    """Particle steps executed on the current GPU since the last reset (soil_hip.h:
    soil_particle_steps) — not in the reference; the numerator of Mparticle-steps/s."""
    n = C.c_uint64(0)
    _call("soil_particle_steps", C.byref(n), 1 if reset else 0, _abi.stream())
    return n.value


def particle_arith(mode=None):
    """Arithmetic of the particle step in the tiled launch shape (soil_hip.h: soil_set_particle_arith;
    not in the reference): "exact" — IEEE quotients, the oracle's walks step for step (default) — or
    "fast" — v_rcp_f32 quotients, statistical parity (tests/test_fast_particles.py).  Returns the mode
    in force; with an argument, sets it first."""
    names = {"exact": 0, "ieee": 0, 0: 0, "fast": 1, 1: 1}
    if mode is not None:
        if mode not in names:
            raise ValueError("particle_arith: 'exact' or 'fast'")
        _call("soil_set_particle_arith", names[mode])
    return "fast" if _abi.lib().soil_get_particle_arith() == 1 else "exact"


def debris_retire(mode=None):
    """What becomes of spent debris walkers in the tiled launch shape (soil_hip.h: soil_set_debris_retire; not
    in the reference): 1 / "on" (default) — a walker whose every further deposit is certain to be an exact
    zero ends its walk; 0 / "off" — every walker is walked to the end as the reference does (same flux planes);
    2 / "watch" — marked and walked on, deposits counted (debris_retire_violations).  Returns the mode in
    force (0, 1, 2); with an argument, sets it first."""
    names = {"off": 0, 0: 0, "on": 1, 1: 1, "watch": 2, 2: 2}
    if mode is not None:
        if mode not in names:
            raise ValueError("debris_retire: 'off', 'on' or 'watch'")
        _call("soil_set_debris_retire", names[mode])
    return _abi.lib().soil_get_debris_retire()


def debris_retire_violations(reset=True):
    """Deposits other than exact zeros made by walkers the watched mode had marked as spent (must be 0)."""
    n = C.c_uint64(0)
    _call("soil_debris_retire_violations", C.byref(n), 1 if reset else 0, _abi.stream())
    return n.value


def layer_merge(height, layers):
    """model.cpp:343-351 -> soil::layer_merge (erosion.cu:747-757)."""
    _call("soil_layer_merge", _f(height, "height"), _f(layers, "layers"), height.elem(),
          _abi.stream())


def albedo_layer(albedo, albedoBedrock, albedoSediment, layers, scaleSediment, shiftSediment):
    """model.cpp:353-369 -> soil::albedo_layer (erosion.cu:877-898)."""
    H, W = _hw(albedo)
    _call("soil_albedo_layer", _f(albedo, "albedo"), _f(albedoBedrock, "albedoBedrock"),
          _f(albedoSediment, "albedoSediment"), _f(layers, "layers"), H * W, float(scaleSediment),
          _abi.vec(shiftSediment, 3), _abi.stream())


def albedo_stratum(albedoBedrock, uplift, layers, scale, param, colorA, colorB, age, freq):
    """model.cpp:371-390 -> soil::albedo_stratum (erosion.cu:828-854)."""
    _call("soil_albedo_stratum", _f(albedoBedrock, "albedoBedrock"), _f(uplift, "uplift"),
          _f(layers, "layers"), uplift.elem(), _abi.vec(scale, 3), param._ref(),
          _abi.vec(colorA, 3), _abi.vec(colorB, 3), float(age), float(freq), _abi.stream())


def albedo_discharge(albedo, discharge, colorDischarge, extinction, scale):
    """model.cpp:393-407 -> soil::albedo_discharge (erosion.cu:900-919)."""
    H, W = _hw(albedo)
    _call("soil_albedo_discharge", _f(albedo, "albedo"), _f(discharge, "discharge"), H * W,
          _abi.vec(colorDischarge, 3), float(extinction), float(scale), _abi.stream())


# -------------------------------------------------------------------- noise

class noise_t:
    """soil::noise_param_t (noise.hpp:14-40), fields bound at model.cpp:413-420."""

    def __init__(self):
        self._c = _abi.NoiseParam()
        _abi.lib().soil_noise_param_default(C.byref(self._c))

    seed = property(lambda s: s._c.seed, lambda s, v: setattr(s._c, "seed", float(v)))
    gain = property(lambda s: s._c.gain, lambda s, v: setattr(s._c, "gain", float(v)))
    lacunarity = property(lambda s: s._c.lacunarity,
                          lambda s, v: setattr(s._c, "lacunarity", float(v)))
    octaves = property(lambda s: s._c.octaves, lambda s, v: setattr(s._c, "octaves", int(v)))
    frequency = property(lambda s: s._c.frequency,
                         lambda s, v: setattr(s._c, "frequency", float(v)))

    @property
    def ext(self):
        return [self._c.ext[0], self._c.ext[1]]

    @ext.setter
    def ext(self, v):
        self._c.ext[0], self._c.ext[1] = float(v[0]), float(v[1])


def noise(shape, param, host=silt.cpu):
    """model.cpp:421 -> soil::noise (noise.hpp:42-56): (H, W) float32 heightmap.
    The reference fills a CPU tensor; `host=silt.gpu` generates straight into HBM
    with the same bits."""
    if not isinstance(shape, silt.shape):
        shape = silt.shape(*shape)
    if shape.dim() != 2:
        raise ValueError("can't extract a full noise buffer from a non-2D index")  # noise.hpp:44-45
    H, W = shape[0], shape[1]
    if host is silt.gpu:
        out = silt.tensor(silt.float32, shape, silt.gpu)
        _call("soil_noise", out.c_ptr, H, W, C.byref(param._c), _abi.stream())
        return out
    dst = np.empty((H, W), np.float32)
    _call("soil_noise_host", dst.ctypes.data_as(C.c_void_p), H, W, C.byref(param._c))
    return silt.tensor._wrap_numpy(dst)


# -------------------------------------------------------------------- timer

s, ms, us, ns = "s", "ms", "us", "ns"  # soil::timer::duration, util.cpp:47-52
_UNIT = {"s": 1.0, "ms": 1e3, "us": 1e6, "ns": 1e9}


class timer:
    """soil::timer (util/timer.hpp:15-69; util.cpp:54-73): wall-clock context
    manager.  Like the reference it does NOT synchronise the device."""

    def __init__(self, duration=ms):
        self._unit = duration
        self._t0 = 0.0
        self.count = 0

    def __enter__(self):
        self._t0 = time.perf_counter()

    def __exit__(self, *exc):
        self.count = int((time.perf_counter() - self._t0) * _UNIT[self._unit])
        return False
