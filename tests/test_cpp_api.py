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
    hip = ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64"] if name == "test_cpp_rccl" else []
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L", libdir, "-lsoil_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"] + hip)
    return exe


def _run(cmd, timeout, env=None):
    """Runs a test binary; on expiry the child is killed and what it printed so far (line-buffered
    MARK lines, tests/cpp/watchdog.hpp) is IN the failure message — a hang names its place."""
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        pytest.fail("%s did not finish in %d s; output so far:\n%s" % (os.path.basename(cmd[0]), timeout, out[-6000:]))
    return proc.returncode, out


def test_cpp_mirror_compiles_and_refuses_without_device(tmp_path):
    from soillib_amd import _abi
    if _abi.lib().soil_device_count() > 0:
        pytest.skip("a HIP device is visible: test_cpp_mirror_on_gpu runs the same binary")
    rc, out = _run([_build(tmp_path)], 120)
    assert rc == 0 and "NO_DEVICE_OK" in out, out


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(tmp_path):
    rc, out = _run([_build(tmp_path)], 90)   # the binary's own watchdog ends it after 60 s
    assert rc == 0 and "CPP_API_OK" in out, out
    _check_against_python_binding(out, ("SLAB0",))


@pytest.mark.gpu
def test_cpp_rccl_wire_on_gpu(tmp_path):
    """The RCCL wire from a bare, torch-less C++ process: slab runner over a
    one-rank communicator, 65.5 MB self-exchange, grouped transfers, all-reduce.  Its own test and
    its own short budget: the round-5 hang of this block took the whole session with it."""
    # the library's own bounded waits first (WireWatch: a transfer that does not complete is aborted and
    # reported), then the binary's watchdog (75 s), then this wrapper (100 s): whichever fires names the place
    env = dict(os.environ, SOIL_RCCL_TIMEOUT_S="15", SOIL_RCCL_INIT_TIMEOUT_S="30", NCCL_DEBUG="WARN")
    rc, out = _run([_build(tmp_path, "test_cpp_rccl")], 100, env=env)
    if rc != 0 and "no return from ncclCommInitRank" in out:
        pytest.skip("RCCL cannot bootstrap a one-rank communicator on this box (the wire itself was not reached):\n"
                    + out[-3000:])
    assert rc == 0 and "CPP_RCCL_OK" in out, out
    if "EXIT_HUNG" in out:
        print("WARNING: every check passed, but the process did not get through the ROCm libraries' exit handlers")
    lib = [l for l in out.splitlines() if l.startswith("RCCL_LIB")]
    assert lib and "librccl" in lib[0], out
    rccl = [l for l in out.splitlines() if l.startswith("RCCL_SELF")]
    assert rccl and rccl[0].split()[2] == "1", out
    assert "RCCL_GROUP ok" in out and "RCCL_REDUCE ok" in out
    print(lib[0])
    print(rccl[0])
    # the runner over RCCL walked the walks of the one-rank wire's runner in test_cpp_api
    rc0, out0 = _run([_build(tmp_path)], 90)
    assert rc0 == 0, out0
    s0 = [l for l in out0.splitlines() if l.startswith("SLAB0")][0].split()
    s1 = [l for l in out.splitlines() if l.startswith("SLAB1")][0].split()
    assert int(s0[1]) == int(s1[1]) and abs(float(s0[2]) - float(s1[2])) <= 1e-6 * abs(float(s0[2]))


@pytest.mark.gpu
def test_rccl_wire_fails_instead_of_hanging(tmp_path):
    """The bounded waits of the RCCL wire on the real library: a rank that never
    shows up ends soil_comm_rccl_create with SOIL_ERR_COMM; a transfer that cannot complete (its stream is
    held) is aborted by the watchdog — ncclCommAbort — and the communicator stays failed, with the
    operation, the byte count and the librccl file in soil_last_error()."""
    exe = _build(tmp_path, "test_cpp_rccl")
    env = dict(os.environ, SOIL_RCCL_INIT_TIMEOUT_S="4", SOIL_RCCL_TIMEOUT_S="2", SOIL_TEST_WATCHDOG_S="50", NCCL_DEBUG="WARN")
    rc, out = _run([exe, "absent"], 70, env=env)
    assert rc == 0 and "CPP_RCCL_OK" in out, out
    line = [l for l in out.splitlines() if l.startswith("RCCL_ABSENT")][0]
    assert "within 4.0 s" in line and 3.5 < float(line.split()[4]) < 15.0, line
    print(line)
    env["SOIL_RCCL_INIT_TIMEOUT_S"] = "45"
    rc, out = _run([exe, "stall"], 70, env=env)
    assert rc == 0 and "CPP_RCCL_OK" in out, out
    line = [l for l in out.splitlines() if l.startswith("RCCL_STALL")][0]
    if "skipped" not in line:
        assert "within 2.0 s" in line and 1.5 < float(line.split()[4]) < 12.0, line
    print(line)


def _check_against_python_binding(stdout, slab_tags):
    # the same three steps through the Python binding of the same step driver (legacy soil.erode)
    import numpy as np
    import soillib as soil
    from soillib_amd import silt
    line = [l for l in stdout.splitlines() if l.startswith("ERODE3")][0].split()
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
    for tag in slab_tags:
        sl = [l for l in stdout.splitlines() if l.startswith(tag)][0].split()
        assert int(sl[1]) == int(line[1])
        assert abs(float(sl[2]) - float(line[2])) <= 1e-6 * abs(h) + 1e-6


def test_cpp_io_roundtrip(tmp_path):
    """soil::io::tiff / geotiff of soil.hpp: host-only, runs without a GPU."""
    rc, out = _run([_build(tmp_path, "test_cpp_io"), str(tmp_path / "cpp.tiff")], 120)
    assert rc == 0 and "CPP_IO_OK" in out, out
