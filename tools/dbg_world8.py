import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from soillib_amd import _abi, silt, soil
from soillib_amd.erosion import ErosionModel
from util import script_param, to_np
lib = _abi.lib()
mode = sys.argv[1]
if mode in ("garbage", "garbage_sub"):
    code = ("import torch; x=[torch.full((1<<28,), 1e30, device='cuda') for _ in range(180)]; torch.cuda.synchronize(); print('filled', len(x))")
    if mode == "garbage_sub":
        print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout)
    else:
        x = [torch.full((1 << 28,), 1e30, device='cuda') for _ in range(150)]
        torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
world, S, W, maxage, steps = [int(v) for v in os.environ.get('DBG_CFG', '8,2048,16384,256,2').split(',')]
sample = [0, 1, S // 2, S - 2, S - 1]
root = '/root/repo'
env = dict(os.environ, SOIL_DEVICE="0", SOIL_DIST_BACKEND="gloo")
for kv in sys.argv[2:]:
    k, v = kv.split("="); env[k] = v
outs = []
for rep in range(1):
    tmp = tempfile.mkdtemp()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29650 + rep),
                          os.path.join(root, "tests", "parallel_gpu_worker.py"), tmp, str(S), str(W), str(maxage), str(steps),
                          ",".join(str(v) for v in sample)], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    parts = [np.load(os.path.join(tmp, "rank%d.npz" % k)) for k in range(world)]
    outs.append(np.concatenate([p["waterHeight"] for p in parts], axis=0))
H = world * S
pp = script_param(soil.param_t()); pp.maxage = maxage
def single():
    m = ErosionModel(H, W, (20.0 / H, 20.0 / W, 4.0), pp, H * W // 8, seed=0)
    npar = soil.noise_t(); npar.seed = 3.0; npar.ext = [H, W]
    bed = soil.noise(silt.shape(H, W), npar, host=silt.gpu)
    zero = silt.tensor(silt.float32, silt.shape(H, W), silt.gpu); silt.set(zero, 0.0)
    _abi.check(lib.soil_layers_from_planes(m.layers.c_ptr, bed.c_ptr, zero.c_ptr, H * W, _abi.stream()))
    silt.set(m.rainfall, 1.0)
    for _ in range(steps): m.step()
    rows = np.array([k * S + r for k in range(world) for r in sample])
    return to_np(m.waterHeight)[rows]
a = single(); b = a; outs.append(outs[0])
def frac(x, y):
    bad = ~(np.isclose(x, y, rtol=1e-4, atol=1e-5 * (np.nanmax(np.abs(y)) + 1e-30)) | (np.isnan(x) & np.isnan(y)))
    return float(bad.mean()), int(np.isnan(x).sum()), int(np.isnan(y).sum())
print(mode, "workers run1 vs run2:", frac(outs[0], outs[1]))
print(mode, "single a vs b:", frac(a, b))
print(mode, "workers1 vs single a:", frac(outs[0], a))
print(mode, "workers2 vs single a:", frac(outs[1], a))
