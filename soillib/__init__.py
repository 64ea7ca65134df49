"""`import soillib as soil` — the reference's Python module name
(python/soillib/__init__.py:1-3: `import silt; from .soillib import *; from .util import *`).

Live API: soillib_amd.soil (mirror of python/source/model.cpp).  Legacy API used
by example/erosion_gpu.py (map_t, data_t, erode, multiply, legacy param names,
normal): soillib_amd.legacy.  IO (tiff, geotiff): soillib_amd.io.  Helpers (`soil.util`):
soillib_amd.util.
"""
import silt  # noqa: F401

from soillib_amd.soil import *  # noqa: F401,F403
from soillib_amd.soil import (accumulate, accumulate_decay, albedo_discharge, albedo_layer,  # noqa: F401
                              albedo_stratum, d4, d8, direction, edge, gaussian_blur, gradient,
                              laplacian, layer_merge, mass_creep, mass_transfer, ms, negslope,
                              noise, noise_t, normal, ns, random_weighted, s, slope,
                              solve_uniform, steepest, timer, transport_debris, transport_fluvial,
                              us)
from soillib_amd.soil import particle_steps  # noqa: F401
from soillib_amd.io import geotiff, geotiff_meta, mesh, tiff  # noqa: F401  (python/source/io.cpp:20-110)
from soillib_amd.legacy import (buffer, clamp, data_t, erode, index, map_t, multiply, param_t,  # noqa: F401
                                resize)
from soillib_amd.silt import cpu, float32, float64, gpu, int32, set, shape, tensor  # noqa: F401,A004  (legacy: soil.float32, soil.gpu, soil.set, ...)
from soillib_amd import util  # noqa: F401
