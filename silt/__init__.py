"""`import silt` — the tensor-runtime names the reference's scripts use
(python/soillib/__init__.py:1), served by soillib_amd.silt."""
from soillib_amd.silt import *  # noqa: F401,F403
from soillib_amd.silt import (add, clone, cpu, float32, float64, gpu, int32, multiply, rng,  # noqa: F401
                              seed, set, shape, tensor, empty_cache)
