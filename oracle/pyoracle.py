"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (soillib_amd/) never
imports this module.  Parity is unpinned by the reference (see soil_oracle.h).

All functions take and return numpy arrays (float32 / int32, C-contiguous)
with the shapes of the reference API: grids (H, W), vec2 planes (H, W, 2),
vec3 planes (H, W, 3).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32P = C.POINTER(C.c_float)
I32P = C.POINTER(C.c_int32)


class Param(C.Structure):
    """soil::param_t, erosion.hpp:17-58 (layout of orc_param / soil_param)."""
    _fields_ = [("maxage", C.c_uint64)] + [(n, C.c_float) for n in (
        "lrate", "timeStep", "exitSlope", "uplift", "rainfall", "gravity", "evapRate",
        "frictionFactor", "fluvialExponent", "suspensionRateFluvial", "depositionRateFluvial",
        "suspensionRateDebris", "depositionRateDebris", "landslideRateDebris",
        "critSlopeBedrock", "critSlopeSediment", "yieldStress", "viscosityWater",
        "bedShearWater", "densityWater", "viscosityDebris", "bedShearDebris",
        "densityDebris")] + [("force", C.c_float * 2), ("_pad", C.c_float)]


class Domain(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("H", "W", "x0", "rows", "r0", "r1")]


class NoiseParam(C.Structure):
    _fields_ = [("frequency", C.c_float), ("octaves", C.c_int32), ("gain", C.c_float),
                ("lacunarity", C.c_float), ("seed", C.c_float), ("ext", C.c_float * 2)]


RNG_DTYPE = np.dtype([("seed", "<u8"), ("offset", "<u8")])


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("soil_oracle.c", "soil_oracle.h", "noise_oracle.c")]
    srcs = [s for s in srcs if os.path.exists(s)]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _declare(l):
    l.orc_expf.restype = C.c_float
    l.orc_expf.argtypes = [C.c_float]
    l.orc_log2f.restype = C.c_float
    l.orc_log2f.argtypes = [C.c_float]
    l.orc_powf.restype = C.c_float
    l.orc_powf.argtypes = [C.c_float, C.c_float]
    l.orc_stepsize.restype = C.c_float
    l.orc_stepsize.argtypes = [C.c_float] * 4
    l.orc_accumulate.restype = C.c_int
    return l


_LIB_FMA = None


class contracted:
    """`with pyoracle.contracted(): ...` — every oracle call inside runs in liboracle_fma.so: the same
    C sources compiled with -ffp-contract=fast -mfma, i.e. with every a*b+c the reference's statements
    spell out fused the way nvcc's default -fmad=true fuses them in the reference's CUDA build.  The
    difference between the two builds is the fp32 tolerance DESIGN.md 4 states against that build."""

    def __enter__(self):
        global _LIB, _LIB_FMA
        lib()
        if _LIB_FMA is None:
            so = os.path.join(_HERE, "liboracle_fma.so")
            srcs = [os.path.join(_HERE, f) for f in ("soil_oracle.c", "soil_oracle.h", "noise_oracle.c")]
            if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs if os.path.exists(x)):
                subprocess.check_call(["make", "-C", _HERE, "liboracle_fma.so"], stdout=subprocess.DEVNULL)
            _LIB_FMA = _declare(C.CDLL(so))
        self._plain = _LIB
        _LIB = _LIB_FMA
        return self

    def __exit__(self, *exc):
        global _LIB
        _LIB = self._plain
        return False


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _declare(C.CDLL(build()))
    return _LIB


def set_threads(n):
    """Threads of the oracle's per-cell loops (the particle loops take `threads=` per call);
    results do not depend on it."""
    lib().orc_set_threads(C.c_int(int(n)))


def default_param():
    p = Param()
    lib().orc_param_default(C.byref(p))
    return p


def domain(H, W, x0=0, rows=None, r0=0, r1=None):
    rows = H if rows is None else rows
    r1 = rows if r1 is None else r1
    return Domain(H, W, x0, rows, r0, r1)


def _f(a):
    return None if a is None else a.ctypes.data_as(F32P)


def _i(a):
    return None if a is None else a.ctypes.data_as(I32P)


def _chk(a, dtype=np.float32):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a


def _scale(s, n):
    arr = (C.c_float * n)(*[float(v) for v in s])
    return arr


# ------------------------------------------------------------------ spec math

def expf(x):
    return np.array([lib().orc_expf(float(v)) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def log2f(x):
    return np.array([lib().orc_log2f(float(v)) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def powf(x, y):
    x, y = np.broadcast_arrays(np.asarray(x, np.float32), np.asarray(y, np.float32))
    return np.array([lib().orc_powf(float(a), float(b)) for a, b in zip(x.ravel(), y.ravel())],
                    np.float32).reshape(x.shape)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def rng_seed(n, seed, offset):
    r = np.zeros(n, RNG_DTYPE)
    r["seed"] = seed
    r["offset"] = offset
    return r


def rng_uniform(rng, subsequence):
    """Draw one uniform from rng[i] with subsequence[i]; advances rng in place."""
    out = np.empty(len(rng), np.float32)
    fn = lib().orc_rng_uniform
    fn.restype = C.c_float
    base = rng.ctypes.data
    for i in range(len(rng)):
        out[i] = fn(C.c_void_p(base + 16 * i), C.c_uint64(int(subsequence[i])))
    return out


def rng_uniform_cell(seed, offset, cells):
    """random_weighted's draw of every cell index in `cells` (block (offset, n >> 2), word n & 3)."""
    fn = lib().orc_rng_uniform_cell
    fn.restype = C.c_float
    return np.array([fn(C.c_uint64(int(seed)), C.c_uint64(int(offset)), C.c_uint64(int(n))) for n in cells],
                    np.float32)


def stepsize(px, py, dx, dy):
    return lib().orc_stepsize(px, py, dx, dy)


def glocal(layers, x, y, scale, exitSlope, dom=None):
    H, W = layers.shape[:2]
    dom = dom or domain(H, W)
    g = (C.c_float * 2)()
    lib().orc_glocal(_f(_chk(layers)), C.byref(dom), _scale(scale, 3), C.c_int64(x), C.c_int64(y),
                     C.c_float(exitSlope), g)
    return np.array([g[0], g[1]], np.float32)


# ---------------------------------------------------------------- erosion ops

def particles_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, rng, layers, waterSource,
                      waterHeight, velocity, albedoSource, scale, param, dom=None, threads=1,
                      remote0=None):
    H, W = (dom.H, dom.W) if dom else layers.shape[:2]
    dom = dom or domain(H, W)
    steps = C.c_int64(0)
    lib().orc_particles_fluvial(
        _f(_chk(waterFlux)), _f(_chk(massFlux)), _f(_chk(velocityFlux)), _f(_chk(albedoFlux)),
        C.c_void_p(rng.ctypes.data), C.c_int64(len(rng)), _f(_chk(layers)), _f(_chk(waterSource)),
        _f(_chk(waterHeight)), _f(_chk(velocity)), _f(_chk(albedoSource)), C.byref(dom),
        _scale(scale, 3), C.byref(param), C.c_int(threads), C.byref(steps), _f(_chk(remote0)))
    return steps.value


def normalize_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, layers, waterSource,
                      waterHeight, mass, velocity, albedoSource, scale, param, dom=None):
    dom = dom or domain(*layers.shape[:2])
    lib().orc_normalize_fluvial(
        _f(_chk(waterFlux)), _f(_chk(massFlux)), _f(_chk(velocityFlux)), _f(_chk(albedoFlux)),
        _f(_chk(layers)), _f(_chk(waterSource)), _f(_chk(waterHeight)), _f(_chk(mass)),
        _f(_chk(velocity)), _f(_chk(albedoSource)), C.byref(dom), _scale(scale, 3),
        C.byref(param))


def particles_debris(massFlux, velocityFlux, albedoFlux, rng, layers, velocity, albedoSource,
                     scale, param, dom=None, threads=1, remote0=None):
    dom = dom or domain(*layers.shape[:2])
    steps = C.c_int64(0)
    lib().orc_particles_debris(
        _f(_chk(massFlux)), _f(_chk(velocityFlux)), _f(_chk(albedoFlux)),
        C.c_void_p(rng.ctypes.data), C.c_int64(len(rng)), _f(_chk(layers)), _f(_chk(velocity)),
        _f(_chk(albedoSource)), C.byref(dom), _scale(scale, 3), C.byref(param),
        C.c_int(threads), C.byref(steps), _f(_chk(remote0)))
    return steps.value


def normalize_debris(massFlux, velocityFlux, albedoFlux, layers, mass, velocity, albedoSource,
                     scale, param, dom=None):
    dom = dom or domain(*layers.shape[:2])
    lib().orc_normalize_debris(
        _f(_chk(massFlux)), _f(_chk(velocityFlux)), _f(_chk(albedoFlux)), _f(_chk(layers)),
        _f(_chk(mass)), _f(_chk(velocity)), _f(_chk(albedoSource)), C.byref(dom),
        _scale(scale, 3), C.byref(param))


def transport_fluvial(layers, rainfall, waterHeight, waterFlux, mass, massFlux, velocity,
                      velocityFlux, albedoFlux, albedoSource, rng, scale, param, threads=1):
    """soil::transport_fluvial, erosion.cu:189-239: particles then normalise."""
    steps = particles_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, rng, layers, rainfall,
                              waterHeight, velocity, albedoSource, scale, param, threads=threads)
    normalize_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, layers, rainfall,
                      waterHeight, mass, velocity, albedoSource, scale, param)
    return steps


def transport_debris(layers, velocity, velocityFlux, mass, massFlux, albedoFlux, albedoSource,
                     rng, scale, param, threads=1):
    """soil::transport_debris, erosion.cu:395-436."""
    steps = particles_debris(massFlux, velocityFlux, albedoFlux, rng, layers, velocity,
                             albedoSource, scale, param, threads=threads)
    normalize_debris(massFlux, velocityFlux, albedoFlux, layers, mass, velocity, albedoSource,
                     scale, param)
    return steps


def mass_transfer(delta, layers, uplift, mass, velocityFluvial, debris, albedo_bedrock,
                  albedoFluxFluvial, albedoFluxDebris, albedo_surface, scale, param, dom=None):
    dom = dom or domain(*layers.shape[:2])
    lib().orc_mass_transfer(
        _f(_chk(delta)), _f(_chk(layers)), _f(_chk(uplift)), _f(_chk(mass)),
        _f(_chk(velocityFluvial)), _f(_chk(debris)), _f(_chk(albedo_bedrock)),
        _f(_chk(albedoFluxFluvial)), _f(_chk(albedoFluxDebris)), _f(_chk(albedo_surface)),
        C.byref(dom), _scale(scale, 3), C.byref(param))


def mass_creep(delta, layers, scale, param, dom=None):
    dom = dom or domain(*layers.shape[:2])
    lib().orc_mass_creep(_f(_chk(delta)), _f(_chk(layers)), C.byref(dom), _scale(scale, 3),
                         C.byref(param))


def layer_merge(layers):
    H, W = layers.shape[:2]
    out = np.empty((H, W), np.float32)
    lib().orc_layer_merge(_f(out), _f(_chk(layers)), C.c_int64(H * W))
    return out


def albedo_stratum(albedoBedrock, uplift, layers, scale, param, colorA, colorB, age, freq):
    lib().orc_albedo_stratum(_f(_chk(albedoBedrock)), _f(_chk(uplift)), _f(_chk(layers)),
                             C.c_int64(uplift.size), _scale(scale, 3), C.byref(param),
                             _scale(colorA, 3), _scale(colorB, 3), C.c_float(age), C.c_float(freq))


def albedo_layer(albedo, albedoBedrock, albedoSediment, layers, scaleSediment, shiftSediment):
    lib().orc_albedo_layer(_f(_chk(albedo)), _f(_chk(albedoBedrock)), _f(_chk(albedoSediment)),
                           _f(_chk(layers)), C.c_int64(layers.size // 2),
                           C.c_float(scaleSediment), _scale(shiftSediment, 3))


def albedo_discharge(albedo, discharge, colorDischarge, extinction, scale):
    lib().orc_albedo_discharge(_f(_chk(albedo)), _f(_chk(discharge)), C.c_int64(discharge.size),
                               _scale(colorDischarge, 3), C.c_float(extinction), C.c_float(scale))


def erode_cells(layers, uplift, rainfall, waterFlux, massFlux, velocityFlux, debrisFlux,
                debrisVelocityFlux, scale, param, dom=None):
    """Reference composition of the cell phase of one step:
    normalize_fluvial, normalize_debris, delta=0, mass_transfer, mass_creep,
    layers += delta, layer_merge.  Returns a dict of the output planes; the
    flux planes are left untouched (the fused kernel re-zeroes them)."""
    dom = dom or domain(*layers.shape[:2])
    rows, W = layers.shape[:2]
    z1 = lambda: np.zeros((rows, W), np.float32)
    z2 = lambda: np.zeros((rows, W, 2), np.float32)
    waterHeight, mass, velocity, debris, debrisVelocity = z1(), z1(), z2(), z1(), z2()
    normalize_fluvial(waterFlux, massFlux, velocityFlux, None, layers, rainfall, waterHeight, mass,
                      velocity, None, scale, param, dom)
    normalize_debris(debrisFlux, debrisVelocityFlux, None, layers, debris, debrisVelocity, None,
                     scale, param, dom)
    delta = z2()
    mass_transfer(delta, layers, uplift, mass, velocity, debris, None, None, None, None, scale,
                  param, dom)
    mass_creep(delta, layers, scale, param, dom)
    layers_next = layers + delta  # silt.add, example/dem_process.py:47
    height = layers_next[..., 0] + layers_next[..., 1]
    return dict(layers_next=layers_next, height=height, waterHeight=waterHeight, mass=mass,
                velocity=velocity, debris=debris, debrisVelocity=debrisVelocity, delta=delta)


# ------------------------------------------------------------------ graph ops

def steepest(height, edge):
    H, W = height.shape
    out = np.empty((H, W), np.int32)
    lib().orc_steepest(_i(out), _f(_chk(height)), C.c_int64(H), C.c_int64(W), C.c_int(edge))
    return out


def direction(height, edge):
    H, W = height.shape
    out = np.empty((H, W), np.int32)
    lib().orc_direction(_i(out), _f(_chk(height)), C.c_int64(H), C.c_int64(W), C.c_int(edge))
    return out


def random_weighted(height, edge, seed, offset, T):
    H, W = height.shape
    out = np.empty((H, W), np.int32)
    lib().orc_random_weighted(_i(out), _f(_chk(height)), C.c_int64(H), C.c_int64(W), C.c_int(edge),
                              C.c_uint64(seed), C.c_uint64(offset), C.c_float(T))
    return out


def slope(tensor, flow, scale):
    H, W = tensor.shape
    out = np.empty((H, W), np.float32)
    lib().orc_slope(_f(out), _f(_chk(tensor)), _i(_chk(flow, np.int32)), C.c_int64(H),
                    C.c_int64(W), _scale(scale, 2))
    return out


def accumulate(graph, source, edge, decay=None):
    H, W = graph.shape
    out = np.empty((H, W), np.float32)
    rc = lib().orc_accumulate(_f(out), _i(_chk(graph, np.int32)), _f(_chk(source)),
                              _f(_chk(decay)), C.c_int64(H), C.c_int64(W), C.c_int(edge))
    assert rc == 0
    return out


# ------------------------------------------------------------------- stencils

def gradient(t, scale):
    H, W = t.shape
    out = np.empty((H, W, 2), np.float32)
    lib().orc_gradient(_f(out), _f(_chk(t)), C.c_int64(H), C.c_int64(W), _scale(scale, 2))
    return out


def negslope(t, scale):
    H, W = t.shape
    out = np.empty((H, W), np.float32)
    lib().orc_negslope(_f(out), _f(_chk(t)), C.c_int64(H), C.c_int64(W), _scale(scale, 2))
    return out


def fill_depressions(height, edge):
    """Priority-flood depression filling (build-defined)."""
    H, W = height.shape
    out = np.empty((H, W), np.float32)
    rc = lib().orc_fill_depressions(_f(out), _f(_chk(height)), C.c_int64(H), C.c_int64(W),
                                    C.c_int(edge))
    assert rc == 0
    return out


def resize(src, newres):
    """soil.resize (multiscale driver): (Ho, Wo[, D]) -> (Hn, Wn[, D])."""
    Ho, Wo = src.shape[:2]
    D = 1 if src.ndim == 2 else src.shape[2]
    Hn, Wn = newres
    out = np.empty((Hn, Wn) + src.shape[2:], np.float32)
    lib().orc_resize(_f(out), _f(_chk(src)), C.c_int64(Hn), C.c_int64(Wn), C.c_int64(Ho),
                     C.c_int64(Wo), C.c_int(D))
    return out


def laplacian(t, scale):
    H, W, D = t.shape
    out = np.empty((H, W, D), np.float32)
    lib().orc_laplacian(_f(out), _f(_chk(t)), C.c_int64(H), C.c_int64(W), C.c_int(D),
                        _scale(scale, 2))
    return out


def gaussian_blur(t, sigma):
    """Blurs a copy (the reference blurs in place and returns its input)."""
    H, W, Cn = t.shape
    t = t.copy()
    scratch = np.empty_like(t)
    lib().orc_gaussian_blur(_f(t), _f(scratch), C.c_int64(H), C.c_int64(W), C.c_int(Cn),
                            C.c_float(sigma))
    return t


def normal(t, scale):
    H, W = t.shape
    out = np.empty((H, W, 3), np.float32)
    lib().orc_normal(_f(out), _f(_chk(t)), C.c_int64(H), C.c_int64(W), _scale(scale, 3))
    return out


def solve_uniform(flow, source, decay, rng, scale, count):
    H, W, K = source.shape
    flux = np.empty((H, W, K), np.float32)
    lib().orc_solve_uniform(_f(flux), _f(_chk(flow)), _f(_chk(source)), _f(_chk(decay)),
                            C.c_void_p(rng.ctypes.data), C.c_int64(len(rng)), C.c_int64(H),
                            C.c_int64(W), C.c_int(K), _scale(scale, 2), C.c_uint64(count))
    return flux


def noise(H, W, frequency=1.0, octaves=8, gain=0.6, lacunarity=2.0, seed=0.0, ext=(512.0, 512.0)):
    p = NoiseParam(frequency, octaves, gain, lacunarity, seed, (C.c_float * 2)(*ext))
    out = np.empty((H, W), np.float32)
    lib().orc_noise(_f(out), C.c_int64(H), C.c_int64(W), C.byref(p))
    return out
