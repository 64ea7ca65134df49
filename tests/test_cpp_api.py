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


def test_cpp_io_roundtrip(tmp_path):
    """soil::io::tiff / geotiff of soil.hpp: host-only, runs without a GPU."""
    exe = _build(tmp_path, "test_cpp_io")
    out = subprocess.run([exe, str(tmp_path / "cpp.tiff")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "CPP_IO_OK" in out.stdout, out.stdout + out.stderr
