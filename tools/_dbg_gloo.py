import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import script_param
from soillib_amd import silt, soil, _abi
from soillib_amd.erosion import ErosionModel
world, S, W, maxage = 2, 96, 128, 24
for steps in (0, 1, 3):
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, SOIL_DEVICE="0", SOIL_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", "29631",
                          os.path.join(ROOT, "tests", "parallel_gpu_worker.py"), tmp, str(S), str(W), str(maxage), str(steps)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    print("steps", steps, "rc", res.returncode, res.stderr[-600:] if res.returncode else "")
    parts = [np.load(os.path.join(tmp, "rank%d.npz" % k)) for k in range(world)]
    got = {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0].files}
    H = world * S
    pp = script_param(soil.param_t()); pp.maxage = maxage
    m = ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), pp, H * W // 8, seed=0)
    npar = soil.noise_t(); npar.seed = 3.0; npar.ext = [H, W]
    bed = soil.noise(silt.shape(H, W), npar, host=silt.gpu)
    _abi.check(_abi.lib().soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, None, bed.elem(), None))
    silt.set(m.rainfall, 1.0)
    for _ in range(steps):
        m.step()
    for k in got:
        want = getattr(m, k).cpu().numpy()
        bad = ~np.isclose(got[k], want, rtol=1e-4, atol=1e-5 * (np.nanmax(np.abs(want)) + 1e-30), equal_nan=True)
        rows = bad.reshape(H, -1).any(axis=1)
        print("  ", k, "bad cells", int(bad.sum()), "bad rows", int(rows.sum()), "first/last bad row",
              (int(np.argmax(rows)), int(H - 1 - np.argmax(rows[::-1]))) if rows.any() else None,
              "got00", got[k].reshape(H, -1)[0, :3], "want00", want.reshape(H, -1)[0, :3])
