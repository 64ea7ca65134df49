"""bench.py --config c1 | c3 — the other BASELINE.json configs in the driver's line format.

  c1  configs[0]: 256x256 GeoTIFF -> normal map on a CPU host tensor (example/tiff_normal.py:12-14).
      A step = soil.geotiff(path) + soil.normal(image.tensor, image.meta.scale).numpy() — the library's
      own TIFF codec and its HOST normal (soil_normal_host); nothing of it runs on the GPU, as in the
      reference.  `value` = Mcells/s of that host path.  The `roofline` block times the DEVICE twin of the
      same operator (soil_normal, 16 algorithmic bytes per cell) on the same tensor.
  c3  configs[2]: 4096x4096 DEM, pit fill, then K = 512 realisations of D8 random_weighted(T = 10, seed 0,
      offset k) + accumulate(rain = 1), averaged (example/dem_multiflow.py:43-49).  A step = one
      realisation; the loop runs inside the library (soil_multiflow) so that no realisation makes the
      trip through host memory the script's `.cpu().numpy()` makes: `value` = H*W*K / t Mcells/s, device
      time only.  `roofline`: one accumulate call (k_donors4 + the rake-compress rounds) timed with HIP
      events, against the bytes that call has to move (profiles/r06_accumulate/algorithmic_bytes.json,
      counted by tools/count_rake_bytes.py from the kernel's own state machine).
(c2 — configs[1], 1024^2 x 10 000 steps — and c4 — configs[3], the default — are bench.py's own step loop.)
"""
import ctypes as C
import json
import os
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0


def _events(abi, n):
    from bench import Events
    return Events(abi, n)


def _base_line(metric, value, unit, steps, warmup, ms, workload, extra_cfg=None):
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": dict({"workload": workload, "parallelism": "single GPU"}, **(extra_cfg or {}))}


# --------------------------------------------------------------------------------------------- c1

def run_c1(args):
    import numpy as np
    import silt
    import soillib as soil
    from soillib_amd import _abi
    lib = _abi.lib()
    H = W = 256
    steps = args.steps if args.steps_given else 200
    warmup = args.warmup if args.warmup_given else 5
    p = soil.noise_t()
    p.seed = 3.0
    p.ext = [H, W]
    height = soil.noise(silt.shape(H, W), p)                 # host tensor: the synthetic DEM
    g = soil.geotiff(height)
    g.meta.scale = [2.0, 2.0, 80.0]
    tmp = tempfile.mkdtemp(prefix="soil_c1_")
    path = os.path.join(tmp, "dem_256.tiff")
    g.write(path)

    def one():
        image = soil.geotiff(path)
        return soil.normal(image.tensor, image.meta.scale).numpy()

    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    t = (time.perf_counter() - t0) / steps
    # the operator alone, host tensor already in memory (what the reference's serial loop is, normal.hpp:25-38)
    image = soil.geotiff(path)
    t0 = time.perf_counter()
    for _ in range(steps):
        soil.normal(image.tensor, image.meta.scale)
    t_op = (time.perf_counter() - t0) / steps
    line = _base_line("Mcells/s, 256^2 GeoTIFF -> normal map on a CPU host tensor", H * W / t / 1e6, "Mcells/s", steps,
                      warmup, t * 1e3,
                      "256x256 float32 GeoTIFF (synthetic OpenSimplex2-FBm DEM, own codec) read + soil.normal on the "
                      "host tensor + .numpy() — example/tiff_normal.py:12-14; no GPU in this path, as in the reference",
                      {"parallelism": "host, one thread (no GPU in this path)", "grid": [H, W], "host_normal_only_ms": t_op * 1e3, "host_normal_only_mcells_per_s": H * W / t_op / 1e6,
                       "threads": 1})
    # the device twin of the operator on the same tensor (inputs resident): what the roofline block prices
    roof = None
    if lib.soil_device_count() > 0:
        dev = image.tensor.gpu()
        ev = _events(_abi, 2)
        for _ in range(10):
            o = soil.normal(dev, image.meta.scale)
        reps = 200
        outs = silt.tensor(silt.float32, silt.shape(H, W, 3), silt.gpu)
        sc = _abi.vec(image.meta.scale, 3)
        ev.record(0)
        for _ in range(reps):
            _abi.check(lib.soil_normal(outs.c_ptr, dev.c_ptr, H, W, sc, _abi.stream()))
        ev.record(1)
        _abi.check(lib.soil_device_synchronize())
        ms = ev.ms(0, 1) / reps
        same = bool((outs.cpu().numpy() == out).all())
        achieved = 16.0 * H * W / (ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_normal (soil_normal, the device twin of the host operator)", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": 16 * H * W, "algorithmic_bytes_per_cell": 16, "avg_launch_ms": ms,
                "device_equals_host_bit_for_bit": same,
                "note": "1 MiB of traffic per launch: %d launches back to back on one stream, so this is the launch "
                        "rate of a 64-work-group kernel, not a bandwidth figure; the 8192^2 figure of the same kernel is "
                        "in profiles/r06_final/bench_stencils.txt" % reps}
    line["roofline"] = roof
    if not args.no_cpu_baseline:
        from oracle import pyoracle as o
        hh = height.numpy()
        o.normal(hh, (2.0, 2.0, 80.0))
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            want = o.normal(hh, (2.0, 2.0, 80.0))
            n += 1
        tc = (time.perf_counter() - t0) / n
        line["cpu_baseline"] = {"value": H * W / tc / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
                                "sample": "the oracle's normal (oracle/soil_oracle.c, scalar C restatement of "
                                          "op/normal.hpp:19-39) on the same 256x256 tensor, %d repetitions, operator "
                                          "only (compare with config.host_normal_only_mcells_per_s)" % n,
                                "equals_product": bool((want == out).all())}
    return line


# --------------------------------------------------------------------------------------------- c3

def run_c3(args):
    import numpy as np
    from soillib_amd import _abi, silt, soil
    lib = _abi.lib()
    S = args.size if args.size_given else 4096
    K = args.steps if args.steps_given else 512
    warm = args.warmup if args.warmup_given else 8
    T = 10.0
    npar = soil.noise_t()
    npar.seed = 3.0
    npar.ext = [S, S]
    h = soil.noise(silt.shape(S, S), npar, host=silt.gpu)
    silt.multiply(h, 100.0)
    rain = silt.tensor(silt.float32, silt.shape(S, S), silt.gpu)
    silt.set(rain, 1.0)
    ev = _events(_abi, 4)
    soil.fill_depressions(h, soil.d8)
    ev.record(0)
    filled = soil.fill_depressions(h, soil.d8)
    ev.record(1)
    _abi.check(lib.soil_device_synchronize())
    fill_ms = ev.ms(0, 1)
    a, b = filled.cpu().numpy(), h.cpu().numpy()
    raised = int((a > b).sum())
    h = filled
    soil.multiflow(h, rain, warm, T)                        # warm-up: workspaces, code objects
    _abi.check(lib.soil_device_synchronize())
    out = silt.tensor(silt.float64, silt.shape(S, S), silt.gpu)
    _abi.check(lib.soil_set_f32(out.c_ptr, 0.0, 2 * S * S, _abi.stream()))
    _abi.check(lib.soil_device_synchronize())
    t0 = time.perf_counter()
    ev.record(0)
    soil.multiflow(h, rain, K, T, out=out)
    ev.record(1)
    _abi.check(lib.soil_device_synchronize())
    wall = time.perf_counter() - t0
    dev_ms = ev.ms(0, 1)
    t = max(wall, dev_ms * 1e-3)
    m = out.cpu().numpy()
    # one accumulate call (the dominant kernels) by itself, HIP events around it on the launch stream
    graph = soil.random_weighted(h, soil.d8, 0, 0, T)
    soil.accumulate(graph, rain, soil.d8)
    reps = 16
    acc_ms = 0.0
    for k in range(reps):
        graph = soil.random_weighted(h, soil.d8, 0, k, T)
        ev.record(2)
        acc = soil.accumulate(graph, rain, soil.d8)       # (synchronises the stream itself, graph.cu:564)
        ev.record(3)
        _abi.check(lib.soil_device_synchronize())
        acc_ms += ev.ms(2, 3)
    acc_ms /= reps
    ev.record(2)
    for k in range(reps):
        graph = soil.random_weighted(h, soil.d8, 0, k, T)
    ev.record(3)
    _abi.check(lib.soil_device_synchronize())
    rw_ms = ev.ms(2, 3) / reps
    alg, alg_note = None, None
    apath = os.path.join(ROOT, "profiles", "r06_accumulate", "algorithmic_bytes.json")
    if os.path.exists(apath):
        try:
            aj = json.load(open(apath))
            if aj.get("grid") == [S, S]:
                alg, alg_note = aj["bytes_per_cell_accumulate"], aj["how"]
        except Exception:
            alg = None
    if alg is None:   # SURVEY 8d's streaming bound: setup ~200 B/cell + 144 B/cell/round, all 2 (ceil(log2(HW)/2) + 1) rounds
        import math
        rounds = 2 * (math.ceil(math.log2(S * S) / 2) + 1)
        alg, alg_note = 200 + 144 * rounds, "SURVEY.md 8d upper bound (every cell pending in every round): not a measured figure"
    achieved = alg * S * S / (acc_ms * 1e-3) / 1e9
    line = _base_line("Mcells/s, %d^2 D8 multi-flow accumulation (random_weighted + accumulate), K = %d realisations" % (S, K),
                      S * S * K / t / 1e6, "Mcells/s", K, warm, t / K * 1e3,
                      "%dx%d DEM (100 x OpenSimplex2-FBm), pit fill, then K = %d realisations of D8 random_weighted(T = 10, seed 0, "
                      "offset k) + accumulate(rain = 1) averaged on the device (soil_multiflow) — example/dem_multiflow.py:43-49 "
                      "without the per-realisation .cpu().numpy()" % (S, S, K),
                      {"grid": [S, S], "K": K, "edge": "D8", "T": T,
                       "fill_depressions_ms": fill_ms, "cells_raised_by_fill": raised,
                       "mean_upstream_area_min_max": [float(m.min()), float(m.max())],
                       "device_ms_per_realisation": dev_ms / K,
                       "one_accumulate_call_ms": acc_ms, "one_random_weighted_call_ms": rw_ms})
    line["roofline"] = {"bound": "hbm", "kernel": "soil_accumulate = k_donors4 + k_rake_compress x rounds (one call)",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "algorithmic_bytes_per_launch": alg * S * S, "algorithmic_bytes_per_cell": alg,
                        "algorithmic_bytes_from": alg_note, "avg_launch_ms": acc_ms}
    tp = os.path.join(ROOT, "profiles", "r06_accumulate", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if tj.get("grid") == [S, S]:
                line["roofline"]["traffic"] = tj["hbm_bytes_per_accumulate"]
        except Exception:
            pass
    if not args.no_cpu_baseline:
        from oracle import pyoracle as o
        cs = 1024
        dem = o.noise(cs, cs, seed=3.0, ext=(float(cs), float(cs))) * np.float32(100.0)
        dem = o.fill_depressions(dem, 1)
        ones = np.ones((cs, cs), np.float32)
        o.set_threads(1)
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 12.0 or n < 2:
            o.accumulate(o.random_weighted(dem, 1, 0, n, T), ones, 1)
            n += 1
        tc = (time.perf_counter() - t0) / n
        line["cpu_baseline"] = {"value": cs * cs / tc / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
                                "sample": "the oracle's random_weighted + accumulate (oracle/soil_oracle.c: the reference's "
                                          "synchronous rake-compress rounds, graph.cu:429-576, scalar C) on a %dx%d DEM of the "
                                          "same kind, %d realisations on one thread" % (cs, cs, n)}
    return line
