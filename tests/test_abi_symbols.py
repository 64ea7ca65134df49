"""The C-ABI library loads and exports every symbol include/soil_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="soil_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(soil_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("soil_transport_fluvial", "soil_transport_debris", "soil_mass_transfer",
                 "soil_mass_creep", "soil_layer_merge", "soil_erode_cells_fused", "soil_steepest",
                 "soil_direction", "soil_random_weighted", "soil_accumulate", "soil_slope",
                 "soil_gradient", "soil_negslope", "soil_laplacian", "soil_gaussian_blur",
                 "soil_normal", "soil_solve_uniform", "soil_noise", "soil_albedo_layer",
                 "soil_albedo_stratum", "soil_albedo_discharge"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from soillib_amd import _abi
    lib = _abi.lib()                       # fails loudly if the .so was never built
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert set(declared_symbols()) == set(_abi.SIGNATURES), \
        set(declared_symbols()) ^ set(_abi.SIGNATURES)
    assert lib.soil_abi_version() == 1
    # include/soil_slab.h: the sharded step (slab runner, communicators)
    slab = declared_symbols("soil_slab.h")
    assert not [s for s in slab if not hasattr(lib, s)]
    assert set(slab) == set(_abi.SLAB_SIGNATURES), set(slab) ^ set(_abi.SLAB_SIGNATURES)


def test_struct_layouts_match_the_header():
    from soillib_amd import _abi
    assert ctypes.sizeof(_abi.Param) == 112      # erosion.hpp:17-58: size_t + 23 float + vec2 (+pad)
    assert ctypes.sizeof(_abi.Rng) == 16
    assert ctypes.sizeof(_abi.Domain) == 48
    assert ctypes.sizeof(_abi.ErosionPlanes) == 15 * 8
    assert ctypes.sizeof(_abi.NoiseParam) == 28
    assert ctypes.sizeof(_abi.Xfer) == 24 and ctypes.sizeof(_abi.Comm) == 56
    assert ctypes.sizeof(_abi.SlabOps) == 8 * 20 and ctypes.sizeof(_abi.SlabConfig) == 80   # (static_assert in slab_runner.hip)
    assert ctypes.sizeof(_abi.SlabInfo) == 9 * 8 + 8 + 16 + 4 * 8 + 16 + 8 + 2 * 8 + 2 * 8 + 8 == 192
    from oracle import pyoracle
    assert ctypes.sizeof(pyoracle.Param) == 112


def test_defaults_match_reference_param_t():
    from soillib_amd import soil
    p = soil.param_t()
    # erosion.hpp:20-56
    assert p.maxage == 512 and p.timeStep == 250.0 and p.exitSlope == pytest.approx(0.02)
    assert p.gravity == pytest.approx(9.81) and p.fluvialExponent == 2.0
    assert p.suspensionRateFluvial == pytest.approx(4.5e-8) and p.critSlopeBedrock == pytest.approx(0.57)
    assert p.bedShearDebris == pytest.approx(0.99) and p.force == [0.0, 0.0]
    with pytest.raises(AttributeError):
        p.samples = 1                      # legacy name: only the shim in soillib accepts it


def test_compute_fails_loudly_without_a_device():
    from soillib_amd import _abi, silt
    lib = _abi.lib()
    if lib.soil_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_abi.SoilError, match="no usable HIP device"):
        silt.tensor(silt.float32, silt.shape(4, 4), silt.gpu)
    rc = lib.soil_layer_merge(None, None, 16, None)
    assert rc == _abi.SOIL_ERR_NO_DEVICE
