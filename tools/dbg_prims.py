import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from soillib_amd import _abi
import test_gpu_parity as T
hip = _abi.lib()
r = np.random.default_rng(3)
x = np.concatenate([-np.exp(r.uniform(-30, 4, 200000)), [0.0, -0.0, -87.0, -100.0, -1e30, np.nan]]).astype(np.float32)
got = T._selftest(hip, x, x, 8).astype(np.float64)
want = np.exp(x.astype(np.float64))
ok = np.isfinite(want) & (want > 1e-37)
err = np.abs(got[ok] / want[ok] - 1.0)
bound = 1.5 * 2.0 ** -23 + np.abs(x[ok].astype(np.float64)) * 2.0 ** -24
i = np.argmax(err / bound)
print('att_exp worst', x[ok][i], got[ok][i], want[ok][i], err[i], bound[i], 'max err for |x|<1', err[np.abs(x[ok]) < 1].max())
f = np.concatenate([r.uniform(-5, 70000, 200000), [0.0, -0.0, -0.25, -1.0, -1.5, 8191.999, 8192.0, 16777215.0, 3e9, -3e9, np.inf, -np.inf, np.nan, 1e-45, -1e-45]]).astype(np.float32)
g = T._selftest(hip, f, f, 9).view(np.int32)
print('floor_cell specials', list(zip(f[-15:].tolist(), g[-15:].tolist())))
want = np.floor(np.nan_to_num(f.astype(np.float64), nan=0.0, posinf=2.0 ** 31 - 1, neginf=-2.0 ** 31))
want = np.clip(want, -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
print('floor_cell mismatches', (g.astype(np.int64) != want).sum())
e = r.integers(127 - 96, 255, 1 << 20).astype(np.uint32)
m = r.integers(0, 1 << 23, 1 << 20, dtype=np.uint32)
v = np.concatenate([((e << 23) | m).view(np.float32), np.array([0.0, np.inf, np.nan, 2.0 ** -96, 1.0, 4.0, 2.0, 3.0], np.float32)])
a = T._selftest(hip, v, v, 10); b = T._selftest(hip, v, v, 11); c = np.sqrt(v)
print('sqrt_rn vs numpy mismatches', (a.view(np.uint32) != c.view(np.uint32)).sum(), 'sqrtf vs numpy', (b.view(np.uint32) != c.view(np.uint32)).sum())
bad = np.nonzero(a.view(np.uint32) != c.view(np.uint32))[0][:5]
print([(v[j], a[j], c[j]) for j in bad])
