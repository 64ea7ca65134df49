"""Multi-process (gloo, CPU) test of the library's slab runner (soil_slab_step,
csrc/slab_runner.hip).

world_size 2 and 3 jobs drive the C++ runner with the oracle plugged in as compute back-end
(soil_slab_ops) and gloo as the wire (soil_comm) — tests/parallel_worker.py; the owned rows of
all ranks, stitched together, must equal a single-domain oracle run of the same global grid:
same trajectories and deposits, fp32 flux summation order aside.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from util import script_param

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _single_domain(oracle, H, W, steps, maxage):
    p = script_param(oracle.default_param())
    p.maxage = maxage
    scale = (20.0 / H, 20.0 / W, 4.0)
    N = H * W // 8
    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0] = oracle.noise(H, W, seed=3.0, ext=(float(H), float(W)))
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    st = dict(layers=layers, waterHeight=z1(), velocity=z2(), debrisVelocity=z2(), debris=z1(),
              height=z1())
    rain = np.ones((H, W), np.float32)
    for step in range(steps):
        rng = oracle.rng_seed(N, 0, step * N)
        wf, mf, vf, df, dvf = z1(), z1(), z2(), z1(), z2()
        oracle.particles_fluvial(wf, mf, vf, None, rng, st["layers"], rain, st["waterHeight"],
                                 st["velocity"], None, scale, p)
        oracle.particles_debris(df, dvf, None, rng, st["layers"], st["debrisVelocity"], None,
                                scale, p)
        r = oracle.erode_cells(st["layers"], z1(), rain, wf, mf, vf, df, dvf, scale, p)
        st = dict(layers=r["layers_next"], waterHeight=r["waterHeight"], velocity=r["velocity"],
                  debrisVelocity=r["debrisVelocity"], debris=r["debris"], height=r["height"])
    return st


@pytest.mark.parametrize("world,S,W,maxage,steps,need,pair", [
    (2, 24, 32, 8, 2, None, False), (3, 16, 24, 6, 2, None, False),
    (2, 80, 48, 48, 4, None, False),   # deep ghost zone (70 rows): the measured reach trims both halos
    (3, 72, 40, 48, 3, "2", False),    # a refresh depth that is too small: launches are repeated
    (2, 80, 48, 48, 3, None, True),    # both launches as one call (soil_particles_pair_slab's slot), trimmed
    (3, 72, 40, 48, 3, "2", True),     # ... and repeated together
])
def test_slab_runner_matches_single_domain(oracle, tmp_path, world, S, W, maxage, steps, need, pair):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        env.pop("SOIL_HALO_NEED", None)
        if need:
            env["SOIL_HALO_NEED"] = need
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "parallel_worker.py"), str(tmp_path),
             str(S), str(W), str(steps), str(maxage)] + (["pair"] if pair else []), env=env,
            stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]

    H = world * S
    want = _single_domain(oracle, H, W, steps, maxage)
    got = {k: [] for k in ("layers", "waterHeight", "velocity", "debris", "height")}
    for rank in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(d["H"]) == H
        G = int(d["G"])
        assert G == int(np.ceil(np.sqrt(2.0) * maxage)) + 2 and G <= S
        for k in got:
            got[k].append(d[k])
        shipped, full = int(d["halo_rows"][0] + d["halo_rows"][1]), int(d["halo_rows"][2])
        if maxage < 16:
            # shallow ghost zone: the margin of the prediction covers it, everything travels, and
            # the ghost rows of the layer plane hold the neighbours' updated rows
            assert shipped <= full and int(d["fallbacks"]) == 0
            x0, rows = int(d["x0"]), int(d["rows"])
            np.testing.assert_allclose(d["ghost_layers"], want["layers"][x0:x0 + rows], rtol=2e-5,
                                       atol=1e-6)
        elif need is None:
            assert shipped < 0.8 * full and int(d["fallbacks"]) == 0, (shipped, full)
            # ... and the launches were given the ghost rows with fresh fields (+ 2), not all G
            assert 0 < int(d["halo_rows"][3]) < 0.9 * int(d["halo_rows"][4]), d["halo_rows"]
        else:
            assert int(d["fallbacks"]) > 0
    for k in got:
        full = np.concatenate(got[k], axis=0)
        w = want[k]
        # includes global cell (0,0): the NaN walkers of every rank reach it through the
        # all-reduced `remote0` buffer, exactly as in the single-domain run
        np.testing.assert_allclose(full, w, rtol=2e-5, atol=1e-6 * (np.nanmax(np.abs(w)) + 1e-30),
                                   err_msg=k)
    assert np.isnan(want["waterHeight"][0, 0])      # the quirk is live on this terrain


def test_slab_layout_partitions_rows():
    from soillib_amd.parallel import slab_layout
    for world, S, G in [(1, 16, 5), (2, 16, 5), (4, 32, 32), (8, 8192, 365)]:
        covered = []
        for r in range(world):
            x0, rows, r0, r1 = slab_layout(r, world, S, G)
            assert r1 - r0 == S and x0 + r0 == r * S
            assert x0 == max(0, r * S - G) and x0 + rows == min(world * S, (r + 1) * S + G)
            covered.extend(range(x0 + r0, x0 + r1))
        assert covered == list(range(world * S))


@pytest.mark.parametrize("world", [2, 3])
def test_multiflow_realisations_sharded_over_ranks(oracle, tmp_path, world):
    """accumulation does not shard, its realisations do — every rank
    ends with the mean over all K realisations of dem_multiflow.py:43-49."""
    H, W, K = 40, 56, 7
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "parallel_worker.py"), "multiflow",
             str(tmp_path), str(H), str(W), str(K)], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    dem = oracle.noise(H, W, seed=1.0, ext=(float(H), float(W))) * 100.0
    rain = np.ones((H, W), np.float32)
    want = np.zeros((H, W), np.float64)
    for k in range(K):                       # the script's serial loop
        want += oracle.accumulate(oracle.random_weighted(dem, 1, 0, k, 10.0), rain, 1) / float(K)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "multiflow_rank%d.npy" % rank))
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)
    assert want.min() >= 1.0 - 1e-6


def test_rccl_bootstrap_failure_falls_back_on_every_rank(tmp_path):
    """default_comm(): when the library's RCCL bootstrap fails (here: no device at all) every
    rank agrees on the gloo wire and says so in describe() — no rank is left waiting in a
    collective the other never enters."""
    script = tmp_path / "fallback.py"
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['SOIL_DIST_BACKEND'] = 'nccl'\n"
        "from soillib_amd import parallel\n"
        "c = parallel.default_comm(device=False)\n"
        "d = c.describe()\n"
        "assert 'FALLBACK' in d['backend'] and d['world_size'] == 2, d\n"
        "c.barrier()\n"
        "print('FALLBACK_OK')\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and out.stdout.count("FALLBACK_OK") == 2, out.stdout + out.stderr


def test_migrate_mode_is_refused_by_a_back_end_that_cannot_hand_walkers_over():
    """SOIL_SLAB_MIGRATE needs soil_slab_ops.particles_pass (the HIP back-end has it; the oracle back-end
    of these CPU jobs does not): soil_slab_create says so instead of running the deep-halo step."""
    from soillib_amd import soil
    from soillib_amd.parallel import CallbackOps, SelfComm, SlabRunner

    class NoPass:                      # every entry NULL: create must stop before it calls any of them
        pass

    p = soil.param_t()
    p.maxage = 8
    with pytest.raises(ValueError, match="particles_pass"):
        SlabRunner(rows_per_rank=32, W=32, param=p, comm=SelfComm(), ops=CallbackOps(NoPass()), mode="migrate")
    with pytest.raises(KeyError):
        SlabRunner(rows_per_rank=32, W=32, param=p, comm=SelfComm(), ops=CallbackOps(NoPass()), mode="sideways")


def test_a_wire_that_never_delivers_fails_the_runner_within_its_timeout(oracle):
    """A dead wire must fail, not hang.  soil_comm_wedged_create blocks in
    every operation like a transfer whose peer never shows up, under the same watchdog as the RCCL
    communicator (WireWatch, csrc/slab_runner.hip 3): the library's runner — oracle plugged in as
    back-end — must come back with SOIL_ERR_COMM naming the operation, and stay failed."""
    import time
    import parallel_worker as pw
    from soillib_amd import _abi, parallel
    param = pw.copy_param(script_param(oracle.default_param()), _abi.Param())
    param.maxage = 8
    comm = parallel.WedgedComm(rank=0, world=2, timeout_s=0.5)
    t0 = time.time()
    with pytest.raises(_abi.CommError) as e:
        runner = parallel.SlabRunner(rows_per_rank=24, W=32, param=param, particles_div=8, seed=0,
                                     ops=parallel.CallbackOps(pw.OracleOps()), comm=comm)
        for _ in range(2):
            runner.step()
    took = time.time() - t0
    msg = str(e.value)
    assert "within 0.5 s" in msg and "wedged test wire" in msg and "aborted" in msg, msg
    assert "bytes" in msg and ("all_reduce_sum_f32" in msg or "exchange" in msg or "barrier" in msg), msg
    assert took < 10.0, took
    # the communicator stays dead: the next call fails at once with the same report
    c = comm.c_comm().contents
    t1 = time.time()
    assert c.barrier(c.ctx) == _abi.SOIL_ERR_COMM and c.status(c.ctx) == _abi.SOIL_ERR_COMM
    assert time.time() - t1 < 0.2 and "within 0.5 s" in _abi.last_error()
    comm.close()
