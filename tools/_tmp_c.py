import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from soillib_amd import _abi, silt, soil
S = 8192
lib = _abi.lib()
for name, D, const in (("rand1", 1, False), ("ones1", 1, True), ("rand2", 2, False), ("ones2", 2, True)):
    t = silt.tensor(silt.float32, silt.shape(S, S, D) if D == 2 else silt.shape(S, S), silt.gpu)
    if const:
        silt.set(t, 1.0)
    else:
        a = np.random.default_rng(1).standard_normal((S, S, D) if D == 2 else (S, S)).astype(np.float32)
        t = silt.tensor.from_numpy(a).gpu()
    for _ in range(3):
        soil.gaussian_blur(t, 3.0)
print(t.cpu().numpy().ravel()[:2])
