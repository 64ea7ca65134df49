"""include/soil.hpp — the C++ host mirror of the reference's `namespace soil` API —
compiles with g++ against the C-ABI library and behaves (GPU) / fails loudly (no GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, name="test_cpp_api"):
    from soillib_amd import _abi
    assert os.path.exists(_abi.LIB_PATH), "build libsoil_hip.so first"
    exe = os.path.join(str(tmp), name)
    libdir = os.path.dirname(_abi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L", libdir, "-lsoil_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_refuses_without_device(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "NO_DEVICE_OK" in out.stdout or "CPP_API_OK" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CPP_API_OK" in out.stdout, out.stdout + out.stderr
    # the grouped ncclSend / ncclRecv path on the real RCCL (a one-rank communicator exchanging with itself)
    rccl = [l for l in out.stdout.splitlines() if l.startswith("RCCL_SELF")]
    assert rccl and rccl[0].split()[2] == "1", out.stdout
    print(rccl[0])
    # the same three steps through the Python binding of the same step driver (legacy soil.erode)
    import numpy as np
    import soillib as soil
    from soillib_amd import silt
    line = [l for l in out.stdout.splitlines() if l.startswith("ERODE3")][0].split()
    S = 96
    q = soil.noise_t()
    q.seed = 3.0
    q.ext = [S, S]
    model = soil.map_t(silt.shape(S, S), [20.0 / S, 20.0 / S, 4.0])
    model.height = soil.noise(silt.shape(S, S), q, host=silt.gpu)
    planes = lambda *dims: silt.tensor(silt.float32, silt.shape(*dims), silt.gpu)
    model.sediment, model.uplift, model.rainfall = planes(S, S), planes(S, S), planes(S, S)
    silt.set(model.sediment, 0.0)
    silt.set(model.uplift, 0.0)
    silt.set(model.rainfall, 1.0)
    data, track = soil.data_t(silt.shape(S, S)), soil.data_t(silt.shape(S, S))
    for d in (data, track):
        d.discharge, d.mass, d.debris = planes(S, S), planes(S, S), planes(S, S)
        d.momentum, d.debris_momentum = planes(S, S, 2), planes(S, S, 2)
        for t in (d.discharge, d.mass, d.debris, d.momentum, d.debris_momentum):
            silt.set(t, 0.0)
    p = soil.param_t()
    p.samples, p.maxage, p.timeStep = S * S // 8, 64, 1000.0
    p.critSlope, p.suspensionRate = 0.57, 0.0008
    soil.particle_steps(reset=True)
    soil.erode(model, data, track, p, 2)
    soil.erode(model, data, track, p)
    assert soil.particle_steps(reset=True) == int(line[1])       # the same walks, step for step
    h = model.height.cpu().numpy().astype(np.float64).sum()
    d = np.nansum(data.discharge.cpu().numpy().astype(np.float64))
    assert abs(h - float(line[2])) <= 1e-6 * abs(h) + 1e-6         # fp32 deposits, summation order
    assert abs(d - float(line[3])) <= 1e-4 * abs(d)
    # the slab runner (soil::slab_runner over soil_slab_*): the same grid, parameters and seed as
    # the three soil::erode steps above, hence the same walks and the same terrain
    for tag in ("SLAB0", "SLAB1"):
        sl = [l for l in out.stdout.splitlines() if l.startswith(tag)][0].split()
        assert int(sl[1]) == int(line[1])
        assert abs(float(sl[2]) - float(line[2])) <= 1e-6 * abs(h) + 1e-6


def test_cpp_io_roundtrip(tmp_path):
    """soil::io::tiff / geotiff of soil.hpp: host-only, runs without a GPU."""
    exe = _build(tmp_path, "test_cpp_io")
    out = subprocess.run([exe, str(tmp_path / "cpp.tiff")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "CPP_IO_OK" in out.stdout, out.stdout + out.stderr
