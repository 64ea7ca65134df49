"""soil.tiff / soil.geotiff (io/tiff.hpp, io/geotiff.hpp; bindings io.cpp:20-100).

The reference reads and writes through libtiff.  Pillow — which links libtiff —
is the independent codec here: files it writes (strips; none / LZW / Deflate /
PackBits; predictors) must decode to the same samples, and files our writer
emits must read back identically in Pillow.  Tiled, big-endian, BigTIFF and
floating-point-predictor files are assembled by a small encoder in this test.
No GPU needed (BASELINE config 1: 256^2 GeoTIFF -> CPU normal map).
"""
import struct
import zlib

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def soil():
    import soillib
    return soillib


@pytest.fixture(scope="module")
def dem():
    r = np.random.default_rng(7)
    x, y = np.meshgrid(np.linspace(0, 4, 96), np.linspace(0, 3, 64))
    return (np.sin(x) * np.cos(y) * 120 + r.standard_normal((64, 96))).astype(np.float32)


def _as_rows(t, h, w):
    """Scanline view of a tensor the reader returns with shape (width, height)."""
    return t.numpy().reshape(-1).reshape(h, w)


# ------------------------------------------------------------- reading Pillow's files

@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
def test_reads_what_libtiff_writes(soil, dem, tmp_path, compression):
    path = str(tmp_path / "pil.tiff")
    PIL.fromarray(dem, mode="F").save(path, compression=compression)
    t = soil.tiff(path)
    assert (t.width, t.height, t.bits) == (96, 64, 32)
    assert tuple(t.shape) == (96, 64)                 # shape(width, height), tiff.hpp:109
    np.testing.assert_array_equal(_as_rows(t.tensor, 64, 96), dem)


def test_reads_horizontal_predictor_and_integers(soil, tmp_path):
    r = np.random.default_rng(3)
    img = (r.integers(0, 60000, (40, 50))).astype(np.uint16)
    path = str(tmp_path / "u16.tiff")
    PIL.fromarray(img).save(path, compression="tiff_lzw",
                            tiffinfo={317: 2})        # Predictor = horizontal differencing
    t = soil.tiff(path)
    assert t.bits == 16 and t.tensor.type.name == "float32"
    np.testing.assert_array_equal(_as_rows(t.tensor, 40, 50), img.astype(np.float32))
    i32 = r.integers(-10**6, 10**6, (17, 23)).astype(np.int32)
    path = str(tmp_path / "i32.tiff")
    PIL.fromarray(i32, mode="I").save(path)
    np.testing.assert_array_equal(_as_rows(soil.tiff(path).tensor, 17, 23), i32.astype(np.float32))


# ------------------------------------------------------------- files built by hand

def _build(rows, *, order="<", big=False, tile=None, compression=1, predictor=1, extra=()):
    """Minimal TIFF encoder: one band of IEEE floats, strips of 7 rows or tiles."""
    h, w = rows.shape
    bps = rows.dtype.itemsize
    e = order

    def encode(block):          # block: 2-D array in file byte order
        raw = np.ascontiguousarray(block.astype(rows.dtype.newbyteorder(e)))
        if predictor == 3 and compression != 1:      # TIFF Technical Note 3
            be = np.ascontiguousarray(block.astype(rows.dtype.newbyteorder(">")))
            planes = be.view(np.uint8).reshape(block.shape[0], block.shape[1], bps)
            rowsb = planes.transpose(0, 2, 1).reshape(block.shape[0], -1)      # MSB plane first
            d = rowsb.copy()
            d[:, 1:] = rowsb[:, 1:] - rowsb[:, :-1]
            data = d.tobytes()
        else:
            data = raw.tobytes()
        if compression == 8:
            data = zlib.compress(data)
        return data

    chunks = []
    if tile:
        tw, th = tile
        for ty in range(0, h, th):
            for tx in range(0, w, tw):
                block = np.zeros((th, tw), rows.dtype)
                part = rows[ty:ty + th, tx:tx + tw]
                block[:part.shape[0], :part.shape[1]] = part
                chunks.append(encode(block))
    else:
        for r0 in range(0, h, 7):
            chunks.append(encode(rows[r0:r0 + 7]))
    off_t, off_f = (16, "Q") if big else (4, "I")
    body = b""
    base = 16 if big else 8
    offsets = []
    for c in chunks:
        offsets.append(base + len(body))
        body += c + (b"\0" if len(c) & 1 else b"")
    tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [8 * bps]), (259, 3, [compression]),
            (262, 3, [1]), (277, 3, [1]), (284, 3, [1]), (317, 3, [predictor]), (339, 3, [3])]
    if tile:
        tags += [(322, 4, [tile[0]]), (323, 4, [tile[1]]), (324, off_t, offsets),
                 (325, off_t, [len(c) for c in chunks])]
    else:
        tags += [(273, off_t, offsets), (278, 4, [7]), (279, off_t, [len(c) for c in chunks])]
    tags += list(extra)
    tags.sort()
    fmt = {2: "s", 3: "H", 4: "I", 12: "d", 16: "Q"}
    ifd_off = base + len(body)
    n_entry = 20 if big else 12
    extra_off = ifd_off + (8 if big else 2) + len(tags) * n_entry + (8 if big else 4)
    ifd = struct.pack(e + ("Q" if big else "H"), len(tags))
    tail = b""
    for tag, typ, vals in tags:
        payload = vals if typ == 2 else struct.pack(e + fmt[typ] * len(vals), *vals)
        count = len(vals)
        ifd += struct.pack(e + "HH" + ("Q" if big else "I"), tag, typ, count)
        inline = 8 if big else 4
        if len(payload) <= inline:
            ifd += payload.ljust(inline, b"\0")
        else:
            if len(tail) & 1:
                tail += b"\0"
            ifd += struct.pack(e + ("Q" if big else "I"), extra_off + len(tail))
            tail += payload
    ifd += struct.pack(e + ("Q" if big else "I"), 0)
    mark = b"II" if e == "<" else b"MM"
    head = (mark + struct.pack(e + "HHHQ", 43, 8, 0, ifd_off)) if big else \
           (mark + struct.pack(e + "HI", 42, ifd_off))
    return head + body + ifd + tail


@pytest.mark.parametrize("kw", [
    dict(), dict(order=">"), dict(big=True), dict(big=True, order=">"),
    dict(tile=(32, 16)), dict(tile=(32, 16), order=">", compression=8),
    dict(compression=8, predictor=3), dict(tile=(16, 16), compression=8, predictor=3, big=True),
    dict(order=">", compression=8, predictor=3), dict(predictor=3),   # no codec, no predictor
])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reads_hand_built_variants(soil, dem, tmp_path, kw, dtype):
    rows = dem.astype(dtype)
    path = tmp_path / "built.tiff"
    path.write_bytes(_build(rows, **kw))
    # Pillow agrees that the file says this (it has no BigTIFF reader, and swaps
    # compressed big-endian floats twice — once in libtiff, once by its raw mode)
    # (nor 64-bit floats)
    if dtype is np.float32 and not kw.get("big") and not (
            kw.get("order") == ">" and kw.get("compression", 1) != 1):
        with PIL.open(str(path)) as im:
            np.testing.assert_array_equal(np.asarray(im), rows)
    t = soil.tiff(str(path))
    assert t.bits == 8 * rows.dtype.itemsize
    assert t.tensor.type.name == ("float64" if dtype is np.float64 else "float32")   # tiff.hpp:116-124
    np.testing.assert_array_equal(_as_rows(t.tensor, *rows.shape), rows)


def test_half_floats_are_converted(soil, tmp_path):
    rows = np.array([[0.0, 1.0, -2.5, 65504.0, 6.1e-5, 6e-8, np.inf]], np.float16)
    path = tmp_path / "f16.tiff"
    path.write_bytes(_build(rows))
    got = _as_rows(soil.tiff(str(path)).tensor, 1, 7)
    np.testing.assert_array_equal(got, rows.astype(np.float32))


def test_missing_or_foreign_file(soil, tmp_path):
    with pytest.raises(FileNotFoundError):            # silt::error::missing_file, tiff.hpp:73
        soil.tiff(str(tmp_path / "nope.tiff"))
    bad = tmp_path / "bad.tiff"
    bad.write_bytes(b"not a tiff at all")
    with pytest.raises(FileNotFoundError):
        soil.geotiff(str(bad))


# ------------------------------------------------------------- writing

def _crafted(kind):
    """Headers a fuzzer would write: every one of them used to crash, hang or abort the process
    (ADVICE r1); libtiff, which the reference reads files with, rejects them with an error."""
    if kind == "bigtiff_entry_count_wraps":      # count * 20 wraps around 2^64
        return b"II" + struct.pack("<HHH", 43, 8, 0) + struct.pack("<Q", 16) + \
            struct.pack("<Q", (1 << 64) // 20 + 1) + b"\0" * 64
    if kind == "long8_count_wraps":              # 8 * 2^61 == 0 mod 2^64: looked like an inline value
        ifd = struct.pack("<Q", 2) + struct.pack("<HHQQ", 256, 16, 1 << 61, 0) + \
            struct.pack("<HHQQ", 257, 4, 1, 4) + struct.pack("<Q", 0)
        return b"II" + struct.pack("<HHH", 43, 8, 0) + struct.pack("<Q", 16) + ifd
    if kind == "unknown_type_huge_count":        # type 99 has no size; 4e9 elements were pushed
        ifd = struct.pack("<H", 3) + struct.pack("<HHII", 256, 99, 0xFFFFFFFF, 0) + \
            struct.pack("<HHII", 257, 3, 1, 4) + struct.pack("<HHII", 273, 99, 0xFFFFFFFF, 8) + \
            struct.pack("<I", 0)
        return b"II" + struct.pack("<HI", 42, 8) + ifd
    if kind == "ifd_offset_near_2_64":           # ifd + head wraps
        return b"II" + struct.pack("<HHH", 43, 8, 0) + struct.pack("<Q", (1 << 64) - 4) + b"\0" * 32
    if kind == "payload_offset_wraps":           # off + bytes wraps
        ifd = struct.pack("<Q", 1) + struct.pack("<HHQQ", 256, 4, 4, (1 << 64) - 8) + struct.pack("<Q", 0)
        return b"II" + struct.pack("<HHH", 43, 8, 0) + struct.pack("<Q", 16) + ifd
    if kind == "giant_tile":                     # a tile of 2^32 x 2^32 samples
        tags = [(256, 4, 1, 16), (257, 4, 1, 16), (258, 3, 1, 32), (259, 3, 1, 8), (339, 3, 1, 3),
                (322, 4, 1, 0xFFFFFFFF), (323, 4, 1, 0xFFFFFFFF), (324, 4, 1, 8), (325, 4, 1, 16)]
        ifd = struct.pack("<H", len(tags)) + b"".join(struct.pack("<HHII", *t) for t in tags) + \
            struct.pack("<I", 0)
        return b"II" + struct.pack("<HI", 42, 8) + ifd
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["bigtiff_entry_count_wraps", "long8_count_wraps",
                                  "unknown_type_huge_count", "ifd_offset_near_2_64",
                                  "payload_offset_wraps", "giant_tile"])
def test_crafted_headers_are_rejected(soil, tmp_path, kind):
    path = tmp_path / (kind + ".tiff")
    path.write_bytes(_crafted(kind))
    for cls in (soil.tiff, soil.geotiff):
        with pytest.raises((RuntimeError, ValueError, OSError)):
            cls(str(path))                        # peeks and reads; must raise, not crash


def test_written_file_is_what_the_reference_emits(soil, dem, tmp_path):
    import silt
    sq = np.ascontiguousarray(dem[:, :64])
    path = str(tmp_path / "out.tiff")
    assert soil.tiff(silt.tensor.from_numpy(sq)).write(path)
    with PIL.open(path) as im:
        tags = im.tag_v2
        assert (tags[256], tags[257], tags[258], tags[259]) == (64, 64, (32,), 1)
        assert tags[262] == 1 and tags[274] == 1 and tags[277] == 1 and tags[284] == 1
        assert tags[339] == (3,) and tags[278] == 64      # IEEEFP; ROWSPERSTRIP = width (tiff.hpp:225)
        np.testing.assert_array_equal(np.asarray(im), sq)
    back = soil.tiff(path)
    np.testing.assert_array_equal(back.tensor.numpy(), sq)
    d = silt.tensor.from_numpy(sq.astype(np.float64))
    soil.tiff(d).write(path)
    assert soil.tiff(path).tensor.type.name == "float64"
    np.testing.assert_array_equal(soil.tiff(path).tensor.numpy(), sq.astype(np.float64))


def test_geotiff_roundtrip_and_nodata(soil, dem, tmp_path):
    import silt
    sq = np.ascontiguousarray(dem[:, :64]).copy()
    sq[3, 5] = -9999.0
    g = soil.geotiff(silt.tensor.from_numpy(sq))
    assert g.meta.coords[3] == 64 and g.meta.coords[4] == 64      # geotiff.hpp:72-73
    g.meta.scale = [30.0, -30.0, 0.0]
    g.meta.coords = [0.0, 0.0, 0.0, 500000.0, 4100000.0, 0.0]
    g.meta.gdal_nodata = "-9999"
    g.meta.gdal_metadata = "<GDALMetadata></GDALMetadata>"
    g.meta.gdal_ascii = "WGS 84 / UTM zone 33N|"
    g.meta.keydir = [1, 1, 0, 1, 1024, 0, 1, 1]
    path = str(tmp_path / "geo.tiff")
    assert g.write(path)
    with PIL.open(path) as im:                        # libtiff sees the GeoTIFF / GDAL tags
        assert im.tag_v2[33550] == (30.0, -30.0, 0.0)
        assert im.tag_v2[33922][3:5] == (500000.0, 4100000.0)
        assert im.tag_v2[42113].rstrip("\0") == "-9999"
        assert tuple(im.tag_v2[34735]) == (1, 1, 0, 1, 1024, 0, 1, 1)
    h = soil.geotiff(path)
    assert h.meta.scale == [30.0, -30.0, 1.0]         # a zero z-scale becomes 1, geotiff.hpp:160-161
    assert h.meta.gdal_nodata == "-9999" and h.meta.gdal_ascii == "WGS 84 / UTM zone 33N|"
    assert h.meta.keydir == [1, 1, 0, 1, 1024, 0, 1, 1]
    a = h.tensor.numpy()
    assert np.isnan(a[3, 5]) and np.isnan(a).sum() == 1          # NoData -> NaN, geotiff.hpp:228-263
    np.testing.assert_array_equal(h.scale, np.array([30.0, -30.0], np.float32))
    np.testing.assert_allclose(h.min, [500000.0, 4100000.0 - 30 * 64])
    np.testing.assert_allclose(h.max, [500000.0 + 30 * 64, 4100000.0])
    np.testing.assert_allclose(h.meta.max, h.max)
    h.unsetnan()                                       # the reference's comparison never matches
    assert np.isnan(h.tensor.numpy()[3, 5])
    h.unsetnan(strict=False)
    assert h.tensor.numpy()[3, 5] == -9999.0
    m = soil.geotiff()
    assert m.peek(path) and (m.width, m.height) == (64, 64) and m.tensor is None   # tiff_merge.py:25-27


def test_config1_geotiff_to_cpu_normal(soil, oracle, tmp_path):
    """BASELINE config 1: a 256^2 GeoTIFF -> soil.normal on the CPU tensor (tiff_normal.py:9-15)."""
    import silt
    H = W = 256
    p = soil.noise_t()
    p.seed = 3.0
    p.ext = [H, W]
    height = soil.noise(silt.shape(H, W), p)
    g = soil.geotiff(height)
    g.meta.scale = [2.0, 2.0, 80.0]
    path = str(tmp_path / "dem_256.tiff")
    g.write(path)
    for file, full in soil.util.iter_tiff(path):
        image = soil.geotiff(full)
        normal = soil.normal(image.tensor, image.meta.scale).numpy()
        assert normal.shape == (H, W, 3)
        want = oracle.normal(height.numpy(), (2.0, 2.0, 80.0))
        np.testing.assert_array_equal(normal, want)
        relief = soil.util.relief_shade(image.tensor.numpy(), normal)
        assert relief.shape == (H, W) and np.isfinite(relief).all()


def test_mesh_ply_export(soil, tmp_path):
    """io/mesh.hpp via example/tiff_mesh.py:15-17: vertices skip NaN cells, quads touching one drop out."""
    import silt
    h = np.arange(12, dtype=np.float32).reshape(3, 4)
    h[1, 2] = np.nan
    m = soil.mesh(silt.tensor.from_numpy(h), [2.0, 3.0, 0.5])
    assert m.vertices.shape == (11, 3) and m.faces.shape == (4, 3)        # 6 quads, 4 touch the NaN... 2 left
    np.testing.assert_array_equal(m.vertices[5], np.array([1 * 2.0, 1 * 3.0, 5 * 0.5], np.float32))
    np.testing.assert_array_equal(m.faces[0], [1, 0, 4])                   # (i01, i00, i10), mesh.hpp:108
    np.testing.assert_array_equal(m.faces[1], [1, 4, 5])                   # (i01, i10, i11)
    lo, hi = m.min.copy(), m.max.copy()
    m.center()
    np.testing.assert_allclose(m.vertices.min(axis=0) + m.vertices.max(axis=0), 0, atol=1e-6)
    path = tmp_path / "mesh.ply"
    assert m.write_binary(str(path))
    raw = path.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert b"format binary_little_endian 1.0" in head and b"element vertex 11" in head
    assert b"element face 4" in head and len(body) == 11 * 12 + 4 * 13
    v = np.frombuffer(body[:11 * 12], "<f4").reshape(11, 3)
    np.testing.assert_array_equal(v, m.vertices)
    assert m.write(str(tmp_path / "mesh_ascii.ply"))
    lines = (tmp_path / "mesh_ascii.ply").read_text().splitlines()
    assert lines[0] == "ply" and lines[1] == "format ascii 1.0" and lines[-1].startswith("3 ")
    assert (hi > lo).all()
