"""ctypes binding of the C ABI in include/soil_hip.h (soillib_amd/lib/libsoil_hip.so).

This is the only place Python touches the shared library.  There is no CPU
fallback: if the library is missing it must be built (soillib_amd.build), and
every compute entry point returns SOIL_ERR_NO_DEVICE without a HIP device,
which `check` turns into a RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOIL_LIB") or os.path.join(_HERE, "lib", "libsoil_hip.so")

SOIL_OK = 0
SOIL_ERR_INVALID_ARGUMENT = -1
SOIL_ERR_NO_DEVICE = -2
SOIL_ERR_HIP = -3
SOIL_ERR_OUT_OF_MEMORY = -4
SOIL_ERR_IO = -5
SOIL_ERR_COMM = -6

D4, D8 = 0, 1
SOIL_CELLS_KEEP_FLUX = 1
SOIL_FLUX_OVERWRITE = 1
SOIL_STEP_FLUX_IN_DIRTY, SOIL_STEP_FLUX_OUT_DIRTY = 1, 2

_PARAM_FLOATS = (
    "lrate", "timeStep", "exitSlope", "uplift", "rainfall", "gravity", "evapRate",
    "frictionFactor", "fluvialExponent", "suspensionRateFluvial", "depositionRateFluvial",
    "suspensionRateDebris", "depositionRateDebris", "landslideRateDebris",
    "critSlopeBedrock", "critSlopeSediment", "yieldStress", "viscosityWater",
    "bedShearWater", "densityWater", "viscosityDebris", "bedShearDebris", "densityDebris")


class Param(C.Structure):
    """soil_param == soil::param_t (erosion.hpp:17-58)."""
    _fields_ = ([("maxage", C.c_uint64)] + [(n, C.c_float) for n in _PARAM_FLOATS] +
                [("force", C.c_float * 2), ("_pad", C.c_float)])


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("offset", C.c_uint64)]


class Domain(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("H", "W", "x0", "rows", "r0", "r1")]


class NoiseParam(C.Structure):
    _fields_ = [("frequency", C.c_float), ("octaves", C.c_int32), ("gain", C.c_float),
                ("lacunarity", C.c_float), ("seed", C.c_float), ("ext", C.c_float * 2)]


_PLANES = ("layers", "layers_next", "height", "uplift", "rainfall", "waterHeight", "waterFlux",
           "mass", "massFlux", "velocity", "velocityFlux", "debris", "debrisFlux",
           "debrisVelocity", "debrisVelocityFlux")


class ErosionPlanes(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PLANES]


_ERODE_MODEL = ("height", "sediment", "uplift", "rainfall", "discharge", "mass", "momentum", "debris",
                "debris_momentum", "discharge_track", "mass_track", "momentum_track", "debris_track",
                "debris_momentum_track")


class ErodeModel(C.Structure):
    """soil_erode_model: the legacy model_t + data + track containers."""
    _fields_ = [(n, C.c_void_p) for n in _ERODE_MODEL]


vp, i64, u64, f32, cint = C.c_void_p, C.c_int64, C.c_uint64, C.c_float, C.c_int
F3 = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/soil_hip.h declares
SIGNATURES = {
    "soil_abi_version": (cint, []),
    "soil_last_error": (C.c_char_p, []),
    "soil_device_count": (cint, []),
    "soil_set_device": (cint, [cint]),
    "soil_device_name": (cint, [C.c_char_p, C.c_size_t]),
    "soil_param_default": (None, [C.POINTER(Param)]),
    "soil_malloc": (cint, [C.POINTER(vp), C.c_size_t]),
    "soil_free": (cint, [vp]),
    "soil_memcpy_h2d": (cint, [vp, vp, C.c_size_t, vp]),
    "soil_memcpy_d2h": (cint, [vp, vp, C.c_size_t, vp]),
    "soil_memcpy_d2d": (cint, [vp, vp, C.c_size_t, vp]),
    "soil_stream_synchronize": (cint, [vp]),
    "soil_device_synchronize": (cint, []),
    "soil_event_create": (cint, [C.POINTER(vp)]),
    "soil_event_destroy": (cint, [vp]),
    "soil_event_record": (cint, [vp, vp]),
    "soil_event_elapsed_ms": (cint, [vp, vp, C.POINTER(f32)]),
    "soil_set_f32": (cint, [vp, f32, i64, vp]),
    "soil_set_i32": (cint, [vp, C.c_int32, i64, vp]),
    "soil_add_f32": (cint, [vp, vp, i64, vp]),
    "soil_multiply_f32": (cint, [vp, f32, i64, vp]),
    "soil_rng_seed": (cint, [vp, i64, u64, u64, vp]),
    "soil_selftest_math": (cint, [vp, vp, vp, i64, cint, vp]),
    "soil_transport_fluvial": (cint, [vp] * 12 + [i64, i64, i64, F3, C.POINTER(Param), vp]),
    "soil_transport_debris": (cint, [vp] * 9 + [i64, i64, i64, F3, C.POINTER(Param), vp]),
    "soil_mass_transfer": (cint, [vp] * 12 + [i64, i64, F3, C.POINTER(Param), vp]),
    "soil_mass_creep": (cint, [vp, vp, i64, i64, F3, C.POINTER(Param), vp]),
    "soil_layer_merge": (cint, [vp, vp, i64, vp]),
    "soil_layers_from_planes": (cint, [vp, vp, vp, i64, vp]),
    "soil_layers_to_planes": (cint, [vp, vp, vp, i64, vp]),
    "soil_albedo_stratum": (cint, [vp, vp, vp, i64, F3, C.POINTER(Param), F3, F3, f32, f32, vp]),
    "soil_albedo_layer": (cint, [vp, vp, vp, vp, i64, f32, F3, vp]),
    "soil_albedo_discharge": (cint, [vp, vp, i64, F3, f32, f32, vp]),
    "soil_erode_cells_fused": (cint, [C.POINTER(ErosionPlanes), C.POINTER(Domain), F3,
                                      C.POINTER(Param), vp]),
    "soil_particles_fluvial_slab": (cint, [vp] * 5 + [i64] + [vp] * 6 +
                                    [C.POINTER(Domain), F3, C.POINTER(Param), vp]),
    "soil_particles_debris_slab": (cint, [vp] * 4 + [i64] + [vp] * 4 +
                                   [C.POINTER(Domain), F3, C.POINTER(Param), vp]),
    "soil_particles_pair_slab": (cint, [C.POINTER(ErosionPlanes), vp, vp, i64, vp,
                                        C.POINTER(Domain), F3, C.POINTER(Param), vp]),
    "soil_particles_pair_slab_ex": (cint, [C.POINTER(ErosionPlanes), vp, vp, i64, vp,
                                           C.POINTER(Domain), F3, C.POINTER(Param), cint, vp]),
    "soil_erode_cells_fused_ex": (cint, [C.POINTER(ErosionPlanes), C.POINTER(Domain), F3,
                                         C.POINTER(Param), cint, vp]),
    "soil_erode_step_ex": (cint, [C.POINTER(ErosionPlanes), vp, i64, u64, u64, i64, i64, F3,
                                  C.POINTER(Param), cint, vp]),
    "soil_erode_step": (cint, [C.POINTER(ErosionPlanes), vp, i64, u64, u64, i64, i64, F3,
                               C.POINTER(Param), vp]),
    "soil_erode": (cint, [C.POINTER(ErodeModel), i64, i64, i64, u64, u64, cint, F3, C.POINTER(Param),
                          vp]),
    "soil_set_particle_mode": (cint, [cint]),
    "soil_set_particle_arith": (cint, [cint]),
    "soil_set_debris_retire": (cint, [cint]),
    "soil_get_debris_retire": (cint, []),
    "soil_debris_retire_violations": (cint, [C.POINTER(u64), cint, vp]),
    "soil_get_particle_arith": (cint, []),
    "soil_ghost_rows": (i64, [C.POINTER(Param)]),
    "soil_particle_steps": (cint, [C.POINTER(u64), cint, vp]),
    "soil_ghost_extent": (cint, [vp, vp, i64, i64, i64, i64, vp]),
    "soil_fill_depressions": (cint, [vp, vp, i64, i64, cint, vp]),
    "soil_multiflow": (cint, [vp, vp, vp, i64, i64, cint, u64, u64, u64, u64, u64, f32, vp]),
    "soil_resize": (cint, [vp, vp, i64, i64, i64, i64, cint, vp]),
    "soil_tiff_peek": (cint, [C.c_char_p, vp]),
    "soil_tiff_tag": (cint, [C.c_char_p, cint, vp, u64, C.POINTER(u64)]),
    "soil_tiff_read": (cint, [C.c_char_p, vp, u64]),
    "soil_tiff_write": (cint, [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "soil_direction": (cint, [vp, vp, i64, i64, cint, vp]),
    "soil_steepest": (cint, [vp, vp, i64, i64, cint, vp]),
    "soil_random_weighted": (cint, [vp, vp, i64, i64, cint, u64, u64, f32, vp]),
    "soil_slope": (cint, [vp, vp, vp, i64, i64, F3, vp]),
    "soil_accumulate": (cint, [vp, vp, vp, vp, i64, i64, cint, vp]),
    "soil_workspace_release": (cint, []),
    "soil_gradient": (cint, [vp, vp, i64, i64, F3, vp]),
    "soil_negslope": (cint, [vp, vp, i64, i64, F3, vp]),
    "soil_laplacian": (cint, [vp, vp, i64, i64, cint, F3, vp]),
    "soil_gaussian_blur": (cint, [vp, vp, i64, i64, cint, f32, vp]),
    "soil_normal": (cint, [vp, vp, i64, i64, F3, vp]),
    "soil_normal_host": (cint, [vp, vp, i64, i64, F3]),
    "soil_solve_uniform": (cint, [vp] * 5 + [i64, i64, i64, cint, F3, u64, vp]),
    "soil_noise_param_default": (None, [C.POINTER(NoiseParam)]),
    "soil_noise": (cint, [vp, i64, i64, C.POINTER(NoiseParam), vp]),
    "soil_noise_host": (cint, [vp, i64, i64, C.POINTER(NoiseParam)]),
    "soil_noise_window": (cint, [vp, i64, i64, i64, C.POINTER(NoiseParam), vp]),
}


# ---- include/soil_slab.h ------------------------------------------------------------

class Xfer(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_int64), ("peer", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_int32, C.POINTER(Xfer), C.c_int32,
                          C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
SOIL_COMM_HOST_ORDERED = 1


class Comm(C.Structure):
    """soil_comm: the wire between the ranks of a slab world."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("flags", C.c_int32),
                ("exchange", EXCHANGE_FN), ("all_reduce_sum_f32", ALLREDUCE_FN), ("barrier", BARRIER_FN),
                ("status", BARRIER_FN)]   # may be NULL: has the wire given up meanwhile?


_PP = C.POINTER(ErosionPlanes)
_DP = C.POINTER(Domain)
_F3P = C.POINTER(C.c_float)
_PARP = C.POINTER(Param)
OPS_FIELDS = [
    ("alloc", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int64)),
    ("release", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)),
    ("fill_f32", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32)),
    ("add_f32", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)),
    ("rng_seed", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64)),
    ("particles_fluvial", C.CFUNCTYPE(C.c_int, C.c_void_p, _PP, C.c_void_p, C.c_int64, C.c_void_p, _DP, _F3P, _PARP)),
    ("particles_debris", C.CFUNCTYPE(C.c_int, C.c_void_p, _PP, C.c_void_p, C.c_int64, C.c_void_p, _DP, _F3P, _PARP)),
    ("particles_pair", C.CFUNCTYPE(C.c_int, C.c_void_p, _PP, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, _DP, _F3P,
                                   _PARP)),
    ("cells", C.CFUNCTYPE(C.c_int, C.c_void_p, _PP, _DP, _F3P, _PARP)),
    ("ghost_extent", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                 C.POINTER(C.c_int32))),
    ("noise_rows", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                               C.POINTER(NoiseParam))),
    ("layers_from_bedrock", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)),
    ("to_host", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)),
    ("from_host", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)),
    ("fork", C.CFUNCTYPE(C.c_int, C.c_void_p)),
    ("join", C.CFUNCTYPE(C.c_int, C.c_void_p)),
    ("sync", C.CFUNCTYPE(C.c_int, C.c_void_p)),
    ("stream", C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32)),
    ("particles_pass", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _PP, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, _DP, _F3P,
                                   _PARP, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)),
]


class SlabOps(C.Structure):
    """soil_slab_ops: the compute back-end of a slab runner."""
    _fields_ = [("ctx", C.c_void_p)] + OPS_FIELDS


class SlabConfig(C.Structure):
    _fields_ = [("rows_per_rank", C.c_int64), ("W", C.c_int64), ("particles_div", C.c_int64),
                ("seed", C.c_uint64), ("scale", C.c_float * 3), ("noise_seed", C.c_float),
                ("noise_rows", C.c_int64), ("init", C.c_int32), ("trim", C.c_int32), ("pair", C.c_int32),
                ("halo_need", C.c_int32), ("mode", C.c_int32)]


class SlabInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("H", "W", "S", "G", "x0", "rows", "r0", "r1", "N")] + [
        ("step_index", C.c_uint64), ("rank", C.c_int32), ("world", C.c_int32), ("trim", C.c_int32),
        ("pair", C.c_int32), ("rows_flux", C.c_int64), ("rows_field", C.c_int64), ("rows_full", C.c_int64),
        ("repeated_launches", C.c_int64), ("reach_hist", C.c_int32 * 4), ("n_reach", C.c_int32),
        ("rows_window", C.c_int64), ("rows_window_full", C.c_int64), ("passes", C.c_int64),
        ("walkers_handed", C.c_int64), ("mode", C.c_int32), ("reserved", C.c_int32)]


MARK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)

SLAB_SIGNATURES = {
    "soil_comm_rccl_unique_id": (cint, [C.POINTER(C.c_uint8)]),
    "soil_comm_rccl_create": (cint, [C.POINTER(C.POINTER(Comm)), C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    "soil_comm_rccl_destroy": (cint, [C.POINTER(Comm)]),
    "soil_comm_rccl_probe": (cint, [C.POINTER(C.c_int32)]),
    "soil_comm_rccl_info": (cint, [C.POINTER(Comm), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32)]),
    "soil_comm_rccl_library": (cint, [C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "soil_comm_wedged_create": (cint, [C.POINTER(C.POINTER(Comm)), C.c_int32, C.c_int32, C.c_double]),
    "soil_comm_wedged_destroy": (cint, [C.POINTER(Comm)]),
    "soil_comm_self_create": (cint, [C.POINTER(C.POINTER(Comm))]),
    "soil_comm_self_destroy": (cint, [C.POINTER(Comm)]),
    "soil_slab_create": (cint, [C.POINTER(vp), C.POINTER(SlabConfig), C.POINTER(Param), C.POINTER(Comm),
                                C.POINTER(SlabOps)]),
    "soil_slab_step": (cint, [vp, MARK_FN, vp]),
    "soil_slab_plane": (cint, [vp, C.c_char_p, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]),
    "soil_slab_get_info": (cint, [vp, C.POINTER(SlabInfo)]),
    "soil_slab_sync": (cint, [vp]),
    "soil_slab_stream": (cint, [vp, C.c_int32, C.POINTER(vp)]),
    "soil_slab_destroy": (cint, [vp]),
    "soil_slab_ops_hip_create": (cint, [C.POINTER(C.POINTER(SlabOps))]),
    "soil_slab_ops_hip_destroy": (cint, [C.POINTER(SlabOps)]),
    "soil_slab_layout": (None, [C.c_int32, C.c_int32, i64, i64, C.POINTER(i64)]),
}

_lib = None


def _load_torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime (torch/lib/libamdhip64.so).
    Two HIP runtimes in one process do not share the device: whichever initialises
    second sees "no GPU" (measured on the MI355X box).  libsoil_hip.so only asks for
    the SONAME libamdhip64.so.7, so when torch is imported FIRST the dynamic linker
    binds it to torch's already-loaded runtime and both live happily on one runtime
    (tensors can then be aliased both ways, RCCL works on our buffers).  torch is
    this package's plumbing for multi-GPU runs anyway, so it goes first whenever it
    is installed; SOIL_NO_TORCH=1 opts out (pure C-ABI use on the system ROCm)."""
    import sys
    if os.environ.get("SOIL_NO_TORCH") == "1" or "torch" in sys.modules:
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """Loads libsoil_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        _load_torch_runtime_first()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsoil_hip.so is missing (%s): build it with `python -m soillib_amd.build`; "
                "there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(SLAB_SIGNATURES.items()):
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def last_error():
    return lib().soil_last_error().decode("utf-8", "replace")


class SoilError(RuntimeError):
    pass


class CommError(SoilError):
    """SOIL_ERR_COMM: the wire between ranks failed or timed out (include/soil_slab.h)."""


def check(rc):
    if rc == SOIL_OK:
        return
    msg = lib().soil_last_error().decode("utf-8", "replace")
    if rc == SOIL_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)  # std::invalid_argument in the reference
    if rc == SOIL_ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    if rc == SOIL_ERR_COMM:
        raise CommError("libsoil_hip error %d: %s" % (rc, msg))
    raise SoilError("libsoil_hip error %d: %s" % (rc, msg))


def vec(values, n):
    values = [float(v) for v in values]
    if len(values) != n:
        raise ValueError("expected %d components, got %d" % (n, len(values)))
    return (C.c_float * n)(*values)


# current stream (hipStream_t as int); 0 = the null stream
_stream = 0


def set_stream(handle):
    """Route subsequent launches to `handle` (e.g. torch.cuda.current_stream().cuda_stream)."""
    global _stream
    _stream = int(handle or 0)


def stream():
    return C.c_void_p(_stream)
