"""One erosion step on one GPU: the build's definition of the legacy `soil.erode`.

The reference snapshot no longer contains `soil::erode` (only its commented-out
binding, python/source/model.cpp:142); what remains are the kernels it was
composed of.  SURVEY.md §3.1 fixes the composition of one step as

    silt.seed(rng, seed, step*N)                       (example/dem_process.py:81)
    transport_fluvial -> transport_debris              (erosion.cu:189-239, :395-436)
    delta = 0; mass_transfer; mass_creep               (erosion.cu:576-611, :712-727)
    layers += delta; layer_merge                       (dem_process.py:47, erosion.cu:747-757)

`ErosionModel.step()` runs exactly that, as two particle kernels followed by
ONE fused cell kernel (soil_erode_cells_fused) that also re-zeroes the flux
planes; `step_unfused()` runs the same step through the individual reference
ops (one launch each) and exists so that tests can show both are bit-identical.
The model may be a row slab of a larger grid (see soillib_amd.parallel).
"""
import ctypes as C
import os

from . import _abi, silt


class ErosionModel:
    """Planes of one erosion model (or one row slab of it), resident in HBM.

    rows    local rows held (owned + ghost), W columns
    dom     _abi.Domain describing where the slab sits in the global grid
    """

    PLANES_1 = ("height", "uplift", "rainfall", "waterHeight", "waterFlux", "mass", "massFlux",
                "debris", "debrisFlux")
    PLANES_2 = ("velocity", "velocityFlux", "debrisVelocity", "debrisVelocityFlux")

    def __init__(self, H, W, scale, param, n_particles, seed=0, dom=None, alloc=None):
        self.H, self.W = int(H), int(W)
        self.scale = [float(v) for v in scale]
        self.param = param
        self.N = int(n_particles)
        self.seed = int(seed)
        self.dom = dom if dom is not None else _abi.Domain(self.H, self.W, 0, self.H, 0, self.H)
        self.rows = int(self.dom.rows)
        self.step_index = 0
        alloc = alloc or (lambda dtype, shape: silt.tensor(dtype, silt.shape(*shape), silt.gpu))
        self._alloc = alloc
        r, w = self.rows, self.W
        self.layers = alloc(silt.float32, (r, w, 2))
        self.layers_next = alloc(silt.float32, (r, w, 2))
        for name in self.PLANES_1:
            setattr(self, name, alloc(silt.float32, (r, w)))
        for name in self.PLANES_2:
            setattr(self, name, alloc(silt.float32, (r, w, 2)))
        self.rng = alloc(silt.rng, (self.N,))
        self.rng_debris = alloc(silt.rng, (self.N,))   # the fluvial launch's state two draws on
        for name in ("layers", "layers_next") + self.PLANES_1 + self.PLANES_2:
            silt.set(getattr(self, name), 0.0)
        silt.seed(self.rng, self.seed, 0)

    # -- helpers -------------------------------------------------------------
    def _planes(self):
        p = _abi.ErosionPlanes()
        for name in _abi._PLANES:
            setattr(p, name, getattr(self, name).ptr)
        return p

    def _scale(self):
        return _abi.vec(self.scale, 3)

    def set_layers(self, layers_tensor):
        """Copy an (rows, W, 2) tensor of (bedrock, sediment) into the model."""
        silt.set(self.layers, layers_tensor)

    # -- the three phases ------------------------------------------------------
    def seed_step(self):
        silt.seed(self.rng, self.seed, self.step_index * self.N)

    def particles_pair(self, overwrite=False):
        """Both particle launches of the step, overlapped (soil_particles_pair_slab).  The
        debris launch draws from its own tensor, seeded where the fluvial launch leaves
        the shared one in the sequential order (two draws per particle further).
        `overwrite`: the flux planes were left as they were by cells_fused(keep_flux=True)
        (SOIL_FLUX_OVERWRITE: the launches' first rounds store instead of adding)."""
        silt.seed(self.rng_debris, self.seed, self.step_index * self.N + 2)
        planes = self._planes()
        _abi.check(_abi.lib().soil_particles_pair_slab_ex(
            C.byref(planes), self.rng.c_ptr, self.rng_debris.c_ptr, self.N, None,
            C.byref(self.dom), self._scale(), self.param._ref(),
            _abi.SOIL_FLUX_OVERWRITE if overwrite else 0, _abi.stream()))

    def particles_fluvial(self):
        L = _abi.lib()
        _abi.check(L.soil_particles_fluvial_slab(
            self.waterFlux.c_ptr, self.massFlux.c_ptr, self.velocityFlux.c_ptr, None,
            self.rng.c_ptr, self.N, self.layers.c_ptr, self.rainfall.c_ptr,
            self.waterHeight.c_ptr, self.velocity.c_ptr, None, None, C.byref(self.dom),
            self._scale(), self.param._ref(), _abi.stream()))

    def particles_debris(self):
        L = _abi.lib()
        _abi.check(L.soil_particles_debris_slab(
            self.debrisFlux.c_ptr, self.debrisVelocityFlux.c_ptr, None, self.rng.c_ptr, self.N,
            self.layers.c_ptr, self.debrisVelocity.c_ptr, None, None, C.byref(self.dom),
            self._scale(), self.param._ref(), _abi.stream()))

    def cells_fused(self, r0=None, r1=None, keep_flux=False):
        """Fused cell phase on local rows [r0, r1) (default: the owned rows).  `keep_flux`: the
        flux planes are not re-zeroed (SOIL_CELLS_KEEP_FLUX) — the next particles_pair must then
        be told to overwrite them."""
        d = self.dom
        dom = _abi.Domain(d.H, d.W, d.x0, d.rows, d.r0 if r0 is None else r0,
                          d.r1 if r1 is None else r1)
        planes = self._planes()
        _abi.check(_abi.lib().soil_erode_cells_fused_ex(
            C.byref(planes), C.byref(dom), self._scale(), self.param._ref(),
            _abi.SOIL_CELLS_KEEP_FLUX if keep_flux else 0, _abi.stream()))

    def swap_layers(self):
        self.layers, self.layers_next = self.layers_next, self.layers

    # -- whole steps -----------------------------------------------------------
    def step(self):
        """One erosion step: 2 particle launches + 1 fused cell launch.  With the whole grid on
        this device it is the library's own step driver (soil_erode_step, csrc/erosion_step.hip:
        seed, both particle launches overlapped on two streams, fused cell phase;
        SOIL_STEP_PAIR=0 in the environment makes it run them one after the other)."""
        if self.rows == self.H:
            planes = self._planes()
            _abi.check(_abi.lib().soil_erode_step(
                C.byref(planes), self.rng.c_ptr, self.N, self.seed, self.step_index, self.H, self.W,
                self._scale(), self.param._ref(), _abi.stream()))
            self.swap_layers()
            self.step_index += 1
            return
        self.seed_step()                       # a slab of a larger grid (soillib_amd.parallel)
        if os.environ.get("SOIL_STEP_PAIR") != "0":
            self.particles_pair()
        else:
            self.particles_fluvial()
            self.particles_debris()
        self.cells_fused()
        self.swap_layers()
        self.step_index += 1

    def step_unfused(self):
        """The same step through the stand-alone reference ops (single GPU only)."""
        from . import soil
        if self.rows != self.H:
            raise ValueError("step_unfused needs the whole grid on one GPU")
        self.seed_step()
        soil.transport_fluvial(self.layers, self.rainfall, self.waterHeight, self.waterFlux,
                               self.mass, self.massFlux, self.velocity, self.velocityFlux, None,
                               None, None, self.rng, self.scale, self.param)
        soil.transport_debris(self.layers, self.debrisVelocity, self.debrisVelocityFlux,
                              self.debris, self.debrisFlux, None, None, None, self.rng, self.scale,
                              self.param)
        delta = self.layers_next  # reuse the spare layer buffer as the delta plane
        silt.set(delta, 0.0)
        soil.mass_transfer(delta, self.layers, self.uplift, self.waterHeight, self.mass,
                           self.velocity, self.debris, self.debrisVelocity, None, None, None, None,
                           self.scale, self.param)
        soil.mass_creep(delta, self.layers, self.scale, self.param)
        silt.add(self.layers, delta)
        soil.layer_merge(self.height, self.layers)
        for name in ("waterFlux", "massFlux", "velocityFlux", "debrisFlux", "debrisVelocityFlux"):
            silt.set(getattr(self, name), 0.0)  # silt.set(track.*, 0)
        self.step_index += 1
