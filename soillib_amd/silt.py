"""silt-lite: the slice of the `silt` tensor runtime the erosion path needs.

The reference takes all of its tensors from `silt` (git@github.com:erosiv/silt,
an empty submodule in the reference snapshot).  This module re-creates the
surface the reference's bindings and example scripts use — names and meaning
inferred from the call sites:

    silt.shape(*dims)  silt.tensor(dtype, shape, host)  tensor.from_numpy / .numpy()
    .cpu() / .gpu()  .shape  .type  .host  silt.float32 / float64 / int32 / rng
    silt.cpu / silt.gpu  silt.set / add / multiply / clone / seed
    (example/erosion_gpu.py:15-26,36; example/dem_multiflow.py:25-31,49;
     example/dem_process.py:27,46-47,72,81,104)

GPU tensors live in HBM (hipMalloc through the C ABI) or alias external device
memory (e.g. a torch tensor) through `tensor.from_device`.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

# ------------------------------------------------------------------ enums


class host_t:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "silt." + self.name


cpu = host_t("cpu")
gpu = host_t("gpu")


class dtype_t:
    def __init__(self, name, np_dtype, itemsize):
        self.name = name
        self.np_dtype = np_dtype
        self.itemsize = itemsize

    def __repr__(self):
        return "silt." + self.name


RNG_NP = np.dtype([("seed", "<u8"), ("offset", "<u8")])

float32 = dtype_t("float32", np.dtype(np.float32), 4)
float64 = dtype_t("float64", np.dtype(np.float64), 8)
int32 = dtype_t("int32", np.dtype(np.int32), 4)
rng = dtype_t("rng", RNG_NP, 16)  # one soil_rng per element (silt::rng = curandState upstream)
int = int32  # noqa: A001  (upstream spells it silt.int)

_BY_NP = {np.dtype(np.float32): float32, np.dtype(np.float64): float64,
          np.dtype(np.int32): int32, RNG_NP: rng}


# ------------------------------------------------------------------ shape

class shape:
    """Dense row-major shape; flatten((x, y)) = x*shape[1] + y."""

    def __init__(self, *dims):
        if len(dims) == 1 and hasattr(dims[0], "__iter__"):
            dims = tuple(dims[0])
        self._dims = tuple(builtins_int(d) for d in dims)
        if not 1 <= len(self._dims) <= 4 or any(d < 0 for d in self._dims):
            raise ValueError("shape needs 1..4 non-negative extents")

    def __getitem__(self, i):
        # indexing past the last axis yields 1, so that shape[2] of an (H, W)
        # tensor reads as a single channel (grad.cu:196, filter.cu:76)
        if isinstance(i, builtins_int) and i >= len(self._dims):
            return 1
        return self._dims[i]

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(self._dims)

    def __eq__(self, other):
        return tuple(self) == tuple(other)

    def __repr__(self):
        return "silt.shape(%s)" % ", ".join(str(d) for d in self._dims)

    def dim(self):
        return len(self._dims)

    def elem(self):
        n = 1
        for d in self._dims:
            n *= d
        return n

    def flatten(self, pos):
        idx = 0
        for p, d in zip(pos, self._dims):
            idx = idx * d + builtins_int(p)
        return idx

    def unflatten(self, index):
        out = []
        for d in reversed(self._dims):
            out.append(index % d)
            index //= d
        return tuple(reversed(out))

    def oob(self, pos):
        return any(p < 0 or p >= d for p, d in zip(pos, self._dims))


import builtins as _builtins  # noqa: E402

builtins_int = _builtins.int


# ----------------------------------------------------------------- tensor

# Freed device blocks are kept for reuse (exact size match) instead of going back to
# the driver: the operator API returns a new tensor per call (as the reference's does),
# and a hipMalloc + hipFree pair costs ~0.25 ms — hipFree also synchronises the device —
# which is more than most per-cell kernels take at 8192^2.  Reuse is stream-ordered: every
# launch of this package goes to the one current stream (_abi.set_stream).
_POOL = {}                      # nbytes -> [device pointers]
_POOL_STATE = {"bytes": 0,
               "limit": builtins_int(os.environ.get("SOIL_POOL_BYTES", str(16 << 30)))}


def empty_cache():
    """Hand the cached device blocks back to the driver."""
    for ptrs in _POOL.values():
        for ptr in ptrs:
            _abi.lib().soil_free(C.c_void_p(ptr))
    _POOL.clear()
    _POOL_STATE["bytes"] = 0


class _DeviceBlock:
    """Owns one hipMalloc'd block; recycled when the last tensor handle dies."""

    def __init__(self, nbytes):
        cached = _POOL.get(nbytes)
        if cached:
            self.ptr = cached.pop()
            _POOL_STATE["bytes"] -= nbytes
        else:
            p = C.c_void_p()
            rc = _abi.lib().soil_malloc(C.byref(p), nbytes)
            if rc != 0 and _POOL_STATE["bytes"]:      # out of memory: drop the cache, try again
                empty_cache()
                rc = _abi.lib().soil_malloc(C.byref(p), nbytes)
            _abi.check(rc)
            self.ptr = p.value or 0
        self.nbytes = nbytes

    def __del__(self):
        try:
            if not self.ptr:
                return
            if self.nbytes and _POOL_STATE["bytes"] + self.nbytes <= _POOL_STATE["limit"]:
                _POOL.setdefault(self.nbytes, []).append(self.ptr)
                _POOL_STATE["bytes"] += self.nbytes
            else:
                _abi.lib().soil_free(C.c_void_p(self.ptr))
        except Exception:
            pass


class tensor:
    """Dynamically typed, ref-counted tensor handle on `silt.cpu` or `silt.gpu`."""

    def __init__(self, dtype, shp, host=cpu):
        if not isinstance(shp, shape):
            shp = shape(*shp) if hasattr(shp, "__iter__") else shape(shp)
        self._dtype = dtype
        self._shape = shp
        self._host = host
        self._np = None
        self._block = None
        self._ptr = 0
        self._keepalive = None
        if host is cpu:
            self._np = np.zeros(tuple(shp), dtype.np_dtype)
        elif host is gpu:
            self._block = _DeviceBlock(shp.elem() * dtype.itemsize)
            self._ptr = self._block.ptr
        else:
            raise ValueError("host must be silt.cpu or silt.gpu")

    # -- construction ------------------------------------------------------
    @classmethod
    def _wrap_numpy(cls, arr):
        t = cls.__new__(cls)
        t._dtype = _BY_NP[arr.dtype]
        t._shape = shape(*arr.shape)
        t._host = cpu
        t._np = arr
        t._block = None
        t._ptr = 0
        t._keepalive = None
        return t

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in _BY_NP:
            raise TypeError("unsupported dtype %s (float32, float64, int32)" % arr.dtype)
        return cls._wrap_numpy(arr.copy())

    @classmethod
    def from_device(cls, ptr, dtype, shp, keepalive=None):
        """Alias device memory owned by someone else (torch tensor, slab view...)."""
        t = cls.__new__(cls)
        t._dtype = dtype
        t._shape = shp if isinstance(shp, shape) else shape(*shp)
        t._host = gpu
        t._np = None
        t._block = None
        t._ptr = builtins_int(ptr)
        t._keepalive = keepalive
        return t

    @classmethod
    def from_torch(cls, t):
        import torch
        kinds = {torch.float32: float32, torch.float64: float64, torch.int32: int32}
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in kinds:
            raise TypeError("from_torch needs a contiguous CUDA float32/float64/int32 tensor")
        return cls.from_device(t.data_ptr(), kinds[t.dtype], shape(*t.shape), keepalive=t)

    # -- properties ----------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def type(self):
        return self._dtype

    @property
    def host(self):
        return self._host

    @property
    def ptr(self):
        """Device pointer (gpu tensors) as an int."""
        if self._host is not gpu:
            raise _abi.SoilError("mismatch_host: expected silt.gpu, got silt.cpu")
        return self._ptr

    @property
    def c_ptr(self):
        return C.c_void_p(self.ptr)

    def elem(self):
        return self._shape.elem()

    def nbytes(self):
        return self.elem() * self._dtype.itemsize

    def __repr__(self):
        return "silt.tensor(%s, %s, %s)" % (self._dtype.name, tuple(self._shape), self._host.name)

    def __setitem__(self, key, value):
        """`t[:] = scalar` or `t[:] = [a, b]` (one value per channel of the trailing axis):
        the fill idiom of the legacy buffers (example/erosion_gpu_multiscale.py:45,63-67)."""
        if key != slice(None):
            raise TypeError("silt.tensor supports only t[:] = value")
        if np.isscalar(value):
            set(self, value)
            return
        v = np.asarray(value, self._dtype.np_dtype).reshape(-1)
        dims = tuple(self._shape)
        if dims[-1] != v.size:
            raise ValueError("t[:] = %r needs a trailing axis of %d" % (value, v.size))
        host = np.ascontiguousarray(np.broadcast_to(v, dims))
        if self._host is cpu:
            self._np[...] = host
        elif host.nbytes:
            _abi.check(_abi.lib().soil_memcpy_h2d(self.c_ptr, host.ctypes.data_as(C.c_void_p),
                                                  host.nbytes, _abi.stream()))
            _abi.check(_abi.lib().soil_stream_synchronize(_abi.stream()))

    # -- movement -------------------------------------------------------------
    def numpy(self):
        if self._host is not cpu:
            raise _abi.SoilError("mismatch_host: numpy() needs a silt.cpu tensor; call .cpu()")
        return self._np

    def cpu(self):
        if self._host is cpu:
            return self
        out = np.empty(tuple(self._shape), self._dtype.np_dtype)
        if out.nbytes:
            _abi.check(_abi.lib().soil_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.c_ptr,
                                                  out.nbytes, _abi.stream()))
        return tensor._wrap_numpy(out)

    def gpu(self):
        if self._host is gpu:
            return self
        out = tensor(self._dtype, self._shape, gpu)
        src = np.ascontiguousarray(self._np)
        if src.nbytes:
            _abi.check(_abi.lib().soil_memcpy_h2d(out.c_ptr, src.ctypes.data_as(C.c_void_p),
                                                  src.nbytes, _abi.stream()))
        return out

    @property
    def __cuda_array_interface__(self):
        if self._host is not gpu or self._dtype is rng:
            raise AttributeError("__cuda_array_interface__")
        return {"shape": tuple(self._shape), "typestr": self._dtype.np_dtype.str,
                "data": (self._ptr, False), "version": 2}

    def view_torch(self):
        """A torch tensor aliasing this gpu tensor's memory (float32/float64/int32)."""
        import torch
        return torch.as_tensor(self, device="cuda")


# -------------------------------------------------------------------- ops

def _same_host(a, b):
    if a.host is not b.host:
        raise _abi.SoilError("mismatch_host: %s vs %s" % (a.host, b.host))


def set(t, value):  # noqa: A001
    """silt.set(tensor, scalar | tensor)  (graph.cu:552-553, erosion_gpu.py:19)."""
    if isinstance(value, tensor):
        _same_host(t, value)
        if t.elem() != value.elem() or t.type is not value.type:
            raise ValueError("silt.set: tensors differ in size or type")
        if t.host is cpu:
            t._np[...] = value._np.reshape(t._np.shape)
        else:
            _abi.check(_abi.lib().soil_memcpy_d2d(t.c_ptr, value.c_ptr, t.nbytes(), _abi.stream()))
        return
    if t.host is cpu:
        t._np[...] = value
    elif t.type is float32:
        _abi.check(_abi.lib().soil_set_f32(t.c_ptr, float(value), t.elem(), _abi.stream()))
    elif t.type is int32:
        _abi.check(_abi.lib().soil_set_i32(t.c_ptr, builtins_int(value), t.elem(), _abi.stream()))
    else:
        raise TypeError("silt.set on the GPU supports float32 and int32")


def add(t, other):
    """silt.add(a, b): a += b  (example/dem_process.py:47)."""
    if isinstance(other, tensor):
        _same_host(t, other)
        if t.elem() != other.elem():
            raise ValueError("silt.add: tensors differ in size")
        if t.host is cpu:
            t._np += other._np.reshape(t._np.shape)
        elif t.type is float32 and other.type is float32:
            _abi.check(_abi.lib().soil_add_f32(t.c_ptr, other.c_ptr, t.elem(), _abi.stream()))
        else:
            raise TypeError("silt.add on the GPU supports float32")
    else:
        if t.host is cpu:
            t._np += other
        else:
            raise TypeError("silt.add(tensor, scalar) on the GPU is not part of the path")


def multiply(t, value):
    """silt.multiply(a, s): a *= s  (example/dem_process.py:46, erosion_gpu.py:14)."""
    if isinstance(value, tensor):
        raise TypeError("silt.multiply(tensor, tensor) is not part of the path")
    if t.host is cpu:
        t._np *= value
    elif t.type is float32:
        _abi.check(_abi.lib().soil_multiply_f32(t.c_ptr, float(value), t.elem(), _abi.stream()))
    else:
        raise TypeError("silt.multiply on the GPU supports float32")


def clone(t):
    """silt.clone(t)  (example/dem_process.py:104)."""
    if t.host is cpu:
        return tensor._wrap_numpy(t._np.copy())
    out = tensor(t.type, t.shape, gpu)
    if t.nbytes():
        _abi.check(_abi.lib().soil_memcpy_d2d(out.c_ptr, t.c_ptr, t.nbytes(), _abi.stream()))
    return out


def seed(t, seed_value, offset):
    """silt.seed(rng, seed, offset): rng[n] <- stream (seed, subsequence n) at `offset`
    (example/dem_process.py:81; curand_init(seed, n, offset), graph.cu:100)."""
    if t.type is not rng:
        raise TypeError("silt.seed needs a silt.rng tensor")
    if t.host is cpu:
        t._np["seed"] = seed_value
        t._np["offset"] = offset
    else:
        _abi.check(_abi.lib().soil_rng_seed(t.c_ptr, t.elem(), builtins_int(seed_value),
                                            builtins_int(offset), _abi.stream()))
