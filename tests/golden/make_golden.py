#!/usr/bin/env python
"""Generates the committed golden fixtures of tests/golden/.  Run in the authoring
container (where /root/reference exists):  python tests/golden/make_golden.py [--accept key1,key2]

Existing fixtures are NOT overwritten unless every key that would change is named in --accept;
the script prints max |delta| per moved key (put that table in the commit message).

  noise_*.npy        soil.noise heightmaps produced by the REFERENCE's own generator:
                     source/soillib/external/FastNoiseLite.h compiled in place into
                     oracle/_ref/libfnl_ref.so (oracle/Makefile target `ref`,
                     oracle/fnl_ref_shim.cpp drives it like op/noise.hpp:14-56).
                     These are the only reference-produced vectors this path has
                     (the reference ships no tests/fixtures).
  oracle_small.npz   inputs and outputs of every oracle op on one small seeded case.
                     Produced by the ORACLE, not by the reference (which cannot be
                     built or imported here): it freezes the oracle's behaviour so
                     that later edits cannot silently change what "parity" means,
                     and lets the GPU tests check the HIP path against data on disk.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as o  # noqa: E402
from util import script_param  # noqa: E402

NOISE_CASES = {
    # name: (H, W, kwargs of soil.noise_t)
    "noise_64x64_seed3": (64, 64, dict(seed=3.0, ext=(64.0, 64.0))),            # erosion_gpu.py:9-15
    "noise_48x80_default": (48, 80, dict(seed=0.0, ext=(512.0, 512.0))),        # noise.hpp defaults
    "noise_33x17_custom": (33, 17, dict(seed=-2.5, ext=(10.0, 7.0), octaves=5, gain=0.45,
                                        lacunarity=2.3, frequency=1.7)),
}


def reference_noise(H, W, frequency=1.0, octaves=8, gain=0.6, lacunarity=2.0, seed=0.0,
                    ext=(512.0, 512.0)):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfnl_ref.so"))
    out = np.empty((H, W), np.float32)
    lib.fnl_ref_noise(out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(H), C.c_int64(W),
                      C.c_float(frequency), C.c_int(octaves), C.c_float(gain),
                      C.c_float(lacunarity), C.c_float(seed), C.c_float(ext[0]),
                      C.c_float(ext[1]))
    return out


def oracle_small():
    H, W, N = 24, 20, 300
    D8 = 1
    r = np.random.default_rng(2026)
    layers = np.zeros((H, W, 2), np.float32)
    layers[..., 0] = o.noise(H, W, seed=3.0, ext=(float(H), float(W)))
    layers[..., 1] = (r.random((H, W)) * 0.02).astype(np.float32)
    h = layers[..., 0].copy()
    scale = (20.0 / H, 20.0 / W, 4.0)
    p = script_param(o.default_param())
    p.maxage = 48
    f1 = lambda s: (r.random((H, W)) * s).astype(np.float32)
    f2 = lambda s: (r.standard_normal((H, W, 2)) * s).astype(np.float32)
    inp = dict(layers=layers, uplift=f1(1.0), rainfall=f1(2.0), waterFlux=f1(3.0),
               massFlux=f1(0.5), velocityFlux=f2(2.0), debrisFlux=f1(0.2),
               debrisVelocityFlux=f2(1.0), velocity=f2(2.0), waterHeight=f1(0.1),
               source=(0.5 + r.random((H, W))).astype(np.float32),
               decay=(0.8 + 0.2 * r.random((H, W))).astype(np.float32),
               blur_in=r.standard_normal((H, W, 2)).astype(np.float32))
    out = {}
    cells = o.erode_cells(layers, inp["uplift"], inp["rainfall"], inp["waterFlux"],
                          inp["massFlux"], inp["velocityFlux"], inp["debrisFlux"],
                          inp["debrisVelocityFlux"], scale, p)
    for k, v in cells.items():
        out["cells_" + k] = v
    z1 = lambda: np.zeros((H, W), np.float32)
    z2 = lambda: np.zeros((H, W, 2), np.float32)
    rng = o.rng_seed(N, 11, 5)
    wf, mf, vf = z1(), z1(), z2()
    steps = o.particles_fluvial(wf, mf, vf, None, rng, layers, inp["rainfall"],
                                inp["waterHeight"], inp["velocity"], None, scale, p)
    out.update(fluvial_waterFlux=wf, fluvial_massFlux=mf, fluvial_velocityFlux=vf,
               fluvial_steps=np.int64(steps))
    df, dvf = z1(), z2()
    pd = script_param(o.default_param())
    pd.maxage = 48
    pd.critSlopeBedrock = 0.05
    pd.yieldStress = 0.001
    o.particles_debris(df, dvf, None, rng, layers, inp["velocity"], None, scale, pd)
    out.update(debris_massFlux=df, debris_velocityFlux=dvf)
    out["steepest_d8"] = o.steepest(h, D8)
    out["steepest_d4"] = o.steepest(h, 0)
    out["direction_d8"] = o.direction(h, D8)
    out["random_weighted_d8"] = o.random_weighted(h, D8, 0, 3, 10.0)
    out["slope"] = o.slope(h, out["steepest_d8"], (0.3, 0.7))
    out["accumulate_d8"] = o.accumulate(out["random_weighted_d8"], inp["source"], D8)
    out["accumulate_decay_d8"] = o.accumulate(out["random_weighted_d8"], inp["source"], D8,
                                              decay=inp["decay"])
    out["gradient"] = o.gradient(h, (0.4, 1.7))
    out["negslope"] = o.negslope(h, (0.4, 1.7))
    out["laplacian2"] = o.laplacian(inp["blur_in"], (0.4, 1.7))
    out["blur2"] = o.gaussian_blur(inp["blur_in"], 3.0)
    out["normal"] = o.normal(h, (0.4, 1.7, 3.0))
    meta = dict(H=H, W=W, N=N, scale=np.array(scale, np.float32), rng_seed=11, rng_offset=5)
    new = {}
    new.update({"in_" + k: v for k, v in inp.items()})
    new.update({"out_" + k: v for k, v in out.items()})
    new.update({"meta_" + k: np.asarray(v) for k, v in meta.items()})
    return new


def diff_report(old, new):
    """[(key, what)] for every key whose stored array would change (bitwise, NaN == NaN)."""
    moved = []
    for k in sorted(set(old) | set(new)):
        if k not in old:
            moved.append((k, "new key"))
        elif k not in new:
            moved.append((k, "key dropped"))
        else:
            a, b = np.asarray(old[k]), np.asarray(new[k])
            if a.shape != b.shape or a.dtype != b.dtype:
                moved.append((k, "shape/dtype %s %s -> %s %s" % (a.shape, a.dtype, b.shape, b.dtype)))
            elif a.tobytes() != b.tobytes():
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                d = d[np.isfinite(d)]
                moved.append((k, "max |delta| %.6g over %d of %d elements" % (
                    d.max() if d.size else float("nan"), int((a != b).sum()), a.size)))
    return moved


def guarded_write(path, new, accept, writer, loader):
    """Fixture discipline: a fixture that is re-cut whenever a kernel wants a
    different oracle freezes nothing.  An existing file is only overwritten when every key that would
    move is named in --accept; the per-key report printed here belongs in the commit message."""
    if os.path.exists(path):
        moved = diff_report(loader(path), new)
        if not moved:
            print("%s: unchanged" % os.path.basename(path))
            return True
        for k, what in moved:
            print("%s: %-28s %s%s" % (os.path.basename(path), k, what, "" if k in accept else "   <-- NOT accepted"))
        refused = [k for k, _ in moved if k not in accept]
        if refused:
            print("REFUSED: %s would change in %d key(s) not listed in --accept (%s); nothing written"
                  % (os.path.basename(path), len(refused), ",".join(refused)))
            return False
    writer(path, new)
    print("%s: written" % os.path.basename(path))
    return True


def main():
    import argparse
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--accept", default="", help="comma-separated keys (oracle_small.npz: in_*/out_*/meta_*; a noise "
                                                 "file: its name) that are allowed to change; everything else must "
                                                 "reproduce bit for bit or the file is left alone")
    args = ap.parse_args()
    accept = set(k for k in args.accept.split(",") if k)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libfnl_ref.so")
    if not os.path.exists(ref_so):
        raise SystemExit("oracle/_ref/libfnl_ref.so missing: run `make -C oracle ref` where "
                         "/root/reference is mounted")
    ok = True
    for name, (H, W, kw) in NOISE_CASES.items():
        ok &= guarded_write(os.path.join(HERE, name + ".npy"), {name: reference_noise(H, W, **kw)}, accept,
                            lambda p, d: np.save(p, next(iter(d.values()))), lambda p: {name: np.load(p)})
    ok &= guarded_write(os.path.join(HERE, "oracle_small.npz"), oracle_small(), accept,
                        lambda p, d: np.savez_compressed(p, **d), lambda p: dict(np.load(p)))
    if not ok:
        raise SystemExit(2)
    print("golden fixtures in", HERE)


if __name__ == "__main__":
    main()
