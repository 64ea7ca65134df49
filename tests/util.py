"""Shared input builders for the parity tests (seeded, deterministic)."""
import numpy as np


def terrain(oracle, H, W, seed=3.0, sediment=0.0, rng_seed=0):
    """(H, W, 2) layer plane: FBm bedrock (soil.noise parameters of
    example/erosion_gpu.py:9-15) + optional random sediment."""
    bed = oracle.noise(H, W, seed=seed, ext=(float(H), float(W)))
    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0] = bed
    if sediment > 0:
        r = np.random.default_rng(rng_seed)
        layers[..., 1] = (r.random((H, W)) * sediment).astype(np.float32)
    return layers


def script_param(oracle_or_param):
    """Parameters of example/erosion_gpu.py:75-100 mapped onto the live names
    (SURVEY.md §8a legacy->live mapping)."""
    p = oracle_or_param
    p.timeStep = 1000.0
    p.maxage = 256
    p.lrate = 1.0
    p.gravity = 9.81
    p.uplift = 0.01
    p.rainfall = 1.0
    p.evapRate = 0.0005
    p.viscosityWater = 0.000001
    p.bedShearWater = 12.5
    p.suspensionRateFluvial = 0.0008
    p.depositionRateFluvial = 0.00001
    p.fluvialExponent = 0.01
    p.exitSlope = 0.025
    p.critSlopeBedrock = 0.57
    p.landslideRateDebris = 0.0025
    p.suspensionRateDebris = 0.00025
    p.depositionRateDebris = 0.0001
    p.yieldStress = 2E6
    p.densityDebris = 2500.0
    p.viscosityDebris = 0.004
    p.bedShearDebris = 60 / 2500.0
    return p


def copy_param(src, dst):
    """Copy a ctypes Param (oracle or product) field by field."""
    for name, _ in src._fields_:
        if name == "force":
            dst.force[0], dst.force[1] = src.force[0], src.force[1]
        elif name != "_pad":
            setattr(dst, name, getattr(src, name))
    return dst


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else a.dtype)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = bits(a) == bits(b)
    both_nan = np.isnan(a) & np.isnan(b) if a.dtype.kind == "f" else np.zeros(a.shape, bool)
    bad = ~(same | both_nan)
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError("%s: %d of %d elements differ, first at %s: %s vs %s" % (
            what, bad.sum(), bad.size, idx.tolist(), a[tuple(idx[0])], b[tuple(idx[0])]))


# ---- device helpers (GPU tests only) ----------------------------------------

def to_gpu(arr):
    from soillib_amd import silt
    return silt.tensor.from_numpy(np.ascontiguousarray(arr)).gpu()


def to_np(t):
    return t.cpu().numpy()


def rng_to_gpu(rng_np):
    """Upload an oracle rng array (structured seed/offset) as a silt.rng tensor."""
    from soillib_amd import silt
    return silt.tensor._wrap_numpy(rng_np.astype(silt.RNG_NP)).gpu()


def product_param(oracle_param):
    """A soillib_amd.soil.param_t carrying the same values as an oracle Param."""
    from soillib_amd import soil
    p = soil.param_t()
    copy_param(oracle_param, p._c)
    return p


# ---- random_weighted: tolerance on the receiver ---------------------------

_D8 = ((-1, 0), (0, -1), (0, 1), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1))   # graph.hpp:21-46


def assert_receivers_close(oracle, got, want, height, K, seed, offset, T, what="random_weighted",
                           max_frac=2e-5, edge_tol=2e-5):
    """The Gibbs weights of `random_weighted` are the reference's fast `__expf` (graph.cu:139): a
    tolerance, and with them every receiver whose draw lies within their error of a CDF edge.  The
    product's map must equal the oracle's but for a counted handful of cells, and each of those must
    be such a cell: its two receivers neighbours in the cumulative order, the draw within `edge_tol`
    of the edge between them (weights recomputed here in float64)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, what
    H, W = got.shape
    bad = np.argwhere(got != want)
    assert len(bad) <= max(1, int(max_frac * got.size)), "%s: %d of %d receivers differ" % (what, len(bad), got.size)
    h = np.asarray(height, np.float64)
    for x, y in bad:
        n = int(x) * W + int(y)
        u = float(oracle.rng_uniform_cell(seed, offset, [n])[0])
        idx, cdf, z = [], [], 0.0
        for k in range(K):
            nx, ny = x + _D8[k][0], y + _D8[k][1]
            if nx < 0 or ny < 0 or nx >= H or ny >= W:
                continue
            dE = (h[x, y] - h[nx, ny]) / (1.0 if k < 4 else float(np.float32(np.sqrt(np.float32(2.0)))))
            z += np.exp(dE / T) if dE > 0 else 0.0
            idx.append(int(nx) * W + int(ny))
            cdf.append(z)
        assert z > 0, "%s: cell (%d, %d) has no downhill neighbour, receivers %d vs %d" % (what, x, y, got[x, y], want[x, y])
        edges = [c / z for c in cdf]
        near = [e for e, i in zip(edges, idx) if abs(u - e) <= edge_tol]
        assert near, "%s: cell (%d, %d): receivers %d vs %d, draw %.9f, edges %s" % (
            what, x, y, got[x, y], want[x, y], u, ["%.9f" % e for e in edges])
        assert got[x, y] in idx + [-1] and want[x, y] in idx + [-1], what


def debris_steps_match(got, want):
    """Steps a debris launch walked against the count of a side that walks every walker to the end (the oracle,
    the direct launch shape): equal — unless this process retires spent debris walkers (soil_set_debris_retire(1);
    the suite runs with them WATCHED, tests/conftest.py: walked to the end, so equal), where it may be fewer."""
    from soillib_amd import soil
    return got <= want if soil.debris_retire() == 1 else got == want
