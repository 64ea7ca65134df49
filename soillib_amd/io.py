"""`soil.tiff`, `soil.geotiff`, `soil.geotiff_meta` — the reference's IO classes
(io/tiff.hpp:20-241, io/geotiff.hpp:63-318) with the surface its Python module
binds (python/source/io.cpp:20-100), over the C ABI's own TIFF codec
(include/soil_hip.h, csrc/io_tiff.hip).  Host-side only; no GPU involved.

Quirks of the reference kept on purpose:
  * a file of `width` x `height` pixels is returned as a tensor of shape
    (width, height) whose memory is the file's scanline order
    (tiff.hpp:109, 135-159) — self-consistent for square rasters only;
  * `geotiff.unsetnan()` compares against NaN with `==` (geotiff.hpp:277-315) and
    therefore never replaces anything; `unsetnan(strict=False)` does what the
    name says.
"""
import ctypes as C

import numpy as np

from . import _abi, silt

_GEO = dict(scale=33550, tiepoints=33922, keydir=34735, params=34736, ascii=34737,
            metadata=42112, nodata=42113)


class _Info(C.Structure):          # soil_tiff_info
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "bits", "sample_format", "samples", "tiled", "tile_width",
        "tile_height", "compression", "predictor", "n_scale", "n_tiepoints", "n_params",
        "n_keydir", "n_ascii", "n_metadata", "n_nodata")]


class _GeoTags(C.Structure):       # soil_geotiff_tags
    _fields_ = [("scale", C.POINTER(C.c_double)), ("n_scale", C.c_uint32),
                ("tiepoints", C.POINTER(C.c_double)), ("n_tiepoints", C.c_uint32),
                ("params", C.POINTER(C.c_double)), ("n_params", C.c_uint32),
                ("keydir", C.POINTER(C.c_int16)), ("n_keydir", C.c_uint32),
                ("ascii", C.c_char_p), ("metadata", C.c_char_p), ("nodata", C.c_char_p)]


def _check_io(rc, filename):
    if rc == -5:                    # SOIL_ERR_IO <- silt::error::missing_file (tiff.hpp:73)
        raise FileNotFoundError(_abi.last_error() or filename)
    _abi.check(rc)


def _fs(filename):
    import os
    return os.fsencode(filename)


def _tag(filename, tag, ctype, count):
    if count == 0:
        return []
    buf = (ctype * count)()
    n = C.c_uint64(0)
    _check_io(_abi.lib().soil_tiff_tag(_fs(filename), tag, buf, C.sizeof(buf), C.byref(n)), filename)
    return list(buf)[: n.value // C.sizeof(ctype)]


def _text(filename, tag, count):
    raw = bytes(bytearray(_tag(filename, tag, C.c_ubyte, count)))
    return raw.split(b"\0", 1)[0].decode("utf-8", "replace")


class tiff:
    """io/tiff.hpp:20-66; `tiff()`, `tiff(filename)` (reads), `tiff(tensor)` (wraps for writing)."""

    def __init__(self, arg=None, *_legacy_index):
        self._width = self._height = self._bits = 0
        self._twidth = self._theight = 0
        self._meta_loaded = False
        self._tiled = False
        self._filename = ""
        self._shape = None
        self._tensor = None
        if isinstance(arg, silt.tensor):           # tiff.hpp:25-39
            self._tensor = arg
            self._shape = arg.shape
            self._width, self._height = arg.shape[0], arg.shape[1]
            if arg.type is silt.float32:
                self._bits = 32
            elif arg.type is silt.float64:
                self._bits = 64
        elif arg is not None:
            self.read(arg)

    # -- tiff.hpp:69-99
    def peek(self, filename):
        info = _Info()
        _check_io(_abi.lib().soil_tiff_peek(_fs(filename), C.byref(info)), filename)
        self._info = info
        self._width, self._height, self._bits = info.width, info.height, info.bits
        self._tiled = bool(info.tiled)
        self._twidth, self._theight = info.tile_width, info.tile_height
        self._filename = str(filename)
        self._meta_loaded = True
        return True

    # -- tiff.hpp:102-213
    def read(self, filename):
        if not self._meta_loaded:
            self.peek(filename)
        dtype = np.float64 if self._bits == 64 else np.float32
        flat = np.empty(self._width * self._height, dtype)
        _check_io(_abi.lib().soil_tiff_read(_fs(filename), flat.ctypes.data_as(C.c_void_p),
                                            flat.nbytes), filename)
        # shape(width, height) over scanline-ordered memory, as the reference builds it
        self._shape = silt.shape(self._width, self._height)
        self._tensor = silt.tensor.from_numpy(flat.reshape(self._width, self._height))
        return True

    # -- tiff.hpp:215-241
    def write(self, filename):
        return self._write(filename, None)

    def _write(self, filename, geo):
        if self._tensor is None or self._bits not in (32, 64):
            raise ValueError("tiff.write: needs a float32 or float64 tensor")
        host = self._tensor.cpu() if self._tensor.host is silt.gpu else self._tensor
        data = np.ascontiguousarray(host.numpy())
        _check_io(_abi.lib().soil_tiff_write(_fs(filename), data.ctypes.data_as(C.c_void_p),
                                             int(self._width), int(self._height),
                                             int(self._bits), geo), filename)
        return True

    bits = property(lambda self: self._bits)
    width = property(lambda self: self._width)
    height = property(lambda self: self._height)
    tensor = property(lambda self: self._tensor)
    shape = property(lambda self: self._shape)


class geotiff_meta:
    """geotiff::meta_t (geotiff.hpp:83-101) as bound in io.cpp:68-98."""

    def __init__(self):
        self.filename = ""
        self.width = self.height = self.bits = 0
        self.gdal_nodata = ""
        self.gdal_metadata = ""
        self.gdal_ascii = ""            # io.cpp:84 binds geoasciiparams under this name
        self.scale = [1.0, 1.0, 1.0]
        self.coords = [0.0] * 6
        self.params = []
        self.keydir = []

    def _origin(self):
        return np.array([self.coords[3], self.coords[4]], np.float32)

    def _far(self):
        s = np.array(self.scale[:2], np.float32)
        return self._origin() + s * np.array([self.width, self.height], np.float32)

    @property
    def min(self):                      # geotiff.hpp:99
        return np.minimum(self._origin(), self._far())

    @min.setter
    def min(self, value):               # io.cpp:90-93
        self.coords[3], self.coords[4] = float(value[0]), float(value[1])

    @property
    def max(self):                      # geotiff.hpp:100
        return np.maximum(self._origin(), self._far())


class geotiff(tiff):
    """io/geotiff.hpp:63-129."""

    def __init__(self, arg=None, *_legacy_index):
        self.meta = geotiff_meta()
        if isinstance(arg, silt.tensor):            # geotiff.hpp:70-74
            tiff.__init__(self, arg)
            self.meta.coords[3] = float(self._shape[0])
            self.meta.coords[4] = float(self._shape[1])
        elif arg is not None:                       # geotiff.hpp:76-79
            tiff.__init__(self)
            self.peek(arg)
            self.read(arg)
        else:
            tiff.__init__(self)

    # -- geotiff.hpp:131-173
    def peek(self, filename):
        tiff.peek(self, filename)
        m, i = self.meta, self._info
        m.filename = self._filename
        m.width, m.height, m.bits = self._width, self._height, self._bits
        if i.n_nodata:
            m.gdal_nodata = _text(filename, _GEO["nodata"], i.n_nodata)
        if i.n_metadata:
            m.gdal_metadata = _text(filename, _GEO["metadata"], i.n_metadata)
        if i.n_ascii:
            m.gdal_ascii = _text(filename, _GEO["ascii"], i.n_ascii)
        if i.n_scale:
            m.scale = _tag(filename, _GEO["scale"], C.c_double, i.n_scale)
            if len(m.scale) > 2 and m.scale[2] == 0.0:     # :160-161
                m.scale[2] = 1.0
        if i.n_tiepoints:
            m.coords = _tag(filename, _GEO["tiepoints"], C.c_double, i.n_tiepoints)
        if i.n_params:
            m.params = _tag(filename, _GEO["params"], C.c_double, i.n_params)
        if i.n_keydir:
            m.keydir = _tag(filename, _GEO["keydir"], C.c_int16, i.n_keydir)
        return True

    # -- geotiff.hpp:175-182
    def read(self, filename):
        self.peek(filename)
        tiff.read(self, filename)
        self._set_nan()
        return True

    # -- geotiff.hpp:183-226
    def write(self, filename):
        m = self.meta
        keep = []

        def arr(values, ctype):
            if not len(values):
                return None, 0
            a = (ctype * len(values))(*values)
            keep.append(a)
            return C.cast(a, C.POINTER(ctype)), len(values)

        g = _GeoTags()
        g.scale, g.n_scale = arr(list(m.scale), C.c_double)
        g.tiepoints, g.n_tiepoints = arr(list(m.coords), C.c_double)
        g.params, g.n_params = arr(list(m.params), C.c_double)
        g.keydir, g.n_keydir = arr([int(v) for v in m.keydir], C.c_int16)
        g.ascii = m.gdal_ascii.encode() if m.gdal_ascii else None
        g.metadata = m.gdal_metadata.encode() if m.gdal_metadata else None
        g.nodata = m.gdal_nodata.encode() if m.gdal_nodata else None
        return self._write(filename, C.byref(g))

    def _nodata_value(self):
        return float(self.meta.gdal_nodata)         # std::stof / std::stod, :232,:243,:254

    # -- geotiff.hpp:228-263
    def _set_nan(self):
        if self.meta.gdal_nodata == "" or self._tensor is None:
            return
        a = self._tensor.numpy()
        a[a == a.dtype.type(self._nodata_value())] = np.nan

    # -- geotiff.hpp:265-315
    def unsetnan(self, strict=True):
        if self.meta.gdal_nodata == "" or self._tensor is None:
            return
        if strict:
            return      # `buffer[i] == nan` is false for every value: the reference changes nothing
        host = self._tensor
        a = host.numpy()
        a[np.isnan(a)] = a.dtype.type(self._nodata_value())

    @property
    def scale(self):                    # geotiff.hpp:105
        return np.array(self.meta.scale[:2], np.float32)

    def _dim(self):
        return np.array([self._width, self._height], np.float32)

    @property
    def min(self):                      # geotiff.hpp:107
        o = np.array([self.meta.coords[3], self.meta.coords[4]], np.float32)
        return np.minimum(o, o + self.scale * self._dim())

    @property
    def max(self):                      # geotiff.hpp:108
        o = np.array([self.meta.coords[3], self.meta.coords[4]], np.float32)
        return np.maximum(o, o + self.scale * self._dim())

    def map(self, p):                   # geotiff.hpp:112
        return self.min + self.scale * np.asarray(p, np.float32)


class mesh:
    """io/mesh.hpp:33-213 — terrain triangle mesh of a height tensor with binary / ASCII
    PLY export (the reference's binding is commented out, io.cpp:104-110; used by
    example/tiff_mesh.py:15-17).  One vertex per non-NaN cell, p = (x, y, value) * scale
    (:62-77); two triangles per 2x2 block whose four cells are all non-NaN (:80-114)."""

    def __init__(self, tensor=None, scale=(1.0, 1.0, 1.0)):
        self.vertices = np.zeros((0, 3), np.float32)
        self.faces = np.zeros((0, 3), np.uint32)
        flt = np.finfo(np.float32)
        self.min = np.full(3, flt.max, np.float32)      # :42-43: max starts at FLT_MIN (> 0),
        self.max = np.full(3, flt.tiny, np.float32)     # as numeric_limits<float>::min() is
        if tensor is not None:
            self._triangulate(tensor, scale)
            if len(self.vertices):
                self.min = np.minimum(self.min, self.vertices.min(axis=0))
                self.max = np.maximum(self.max, self.vertices.max(axis=0))

    def _triangulate(self, tensor, scale):
        host = tensor.cpu() if tensor.host is silt.gpu else tensor
        a = host.numpy()
        H, W = a.shape[0], a.shape[1]
        a = a.reshape(H, W)
        ok = ~np.isnan(a)
        ids = np.full(H * W, -1, np.int64)
        ids[ok.reshape(-1)] = np.arange(int(ok.sum()))
        ids = ids.reshape(H, W)
        x, y = np.nonzero(ok)                           # flat (row-major) order, :63-77
        s = np.asarray(scale, np.float32)
        self.vertices = (np.stack([x, y, a[ok]], axis=1).astype(np.float32) * s).astype(np.float32)
        quad = ok[:-1, :-1] & ok[:-1, 1:] & ok[1:, :-1] & ok[1:, 1:]
        i00, i01 = ids[:-1, :-1][quad], ids[:-1, 1:][quad]
        i10, i11 = ids[1:, :-1][quad], ids[1:, 1:][quad]
        f0 = np.stack([i01, i00, i10], axis=1)          # :108-109
        f1 = np.stack([i01, i10, i11], axis=1)
        self.faces = np.stack([f0, f1], axis=1).reshape(-1, 3).astype(np.uint32)

    def center(self):                                   # :117-122
        self.vertices = self.vertices - np.float32(0.5) * (self.max + self.min)

    def write(self, filename):                          # :134-168: ASCII, positions normalised
        with open(filename, "w") as out:
            out.write("ply\nformat ascii 1.0\ncomment Created in soillib\n")
            out.write("element vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                      % len(self.vertices))
            out.write("element face %d\nproperty list uchar uint vertex_indices\nend_header\n"
                      % len(self.faces))
            with np.errstate(divide="ignore", invalid="ignore"):
                vm = (self.vertices - self.min) / (self.max - self.min)
            for v in vm:
                out.write("%g %g %g\n" % (v[0], v[1], v[2]))
            for f in self.faces:
                out.write("3 %d %d %d\n" % (f[0], f[1], f[2]))
        return True

    def write_binary(self, filename):                   # :170-207
        with open(filename, "wb") as out:
            out.write(b"ply\nformat binary_little_endian 1.0\n")
            out.write(b"element vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                      % len(self.vertices))
            out.write(b"element face %d\nproperty list uchar uint vertex_indices\nend_header\n"
                      % len(self.faces))
            out.write(np.ascontiguousarray(self.vertices, "<f4").tobytes())
            rec = np.zeros(len(self.faces), np.dtype([("n", "u1"), ("v", "<u4", 3)]))
            rec["n"] = 3
            rec["v"] = self.faces
            out.write(rec.tobytes())
        return True
