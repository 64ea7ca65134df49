#!/usr/bin/env python
"""A/B of k_erode_cells_fused variants INSIDE the bench step (after the particle phase, planes where
the library's pool puts them), alternating variants in one process on one box.

    python tools/ab_cells_in_bench.py [--size 8192] [--rounds 3] [--steps 6] [--variants 0,4]

SOIL_CELLS_VARIANT is read by the library on every launch: 0 = product (two-channel stores swapped
through LDS into 1 KiB-contiguous instructions), 4 = round 1's direct stores, 1/3 = 512/128-thread
work-groups, 2 = no XCD remap.  Prints the cell-phase time per variant and round, and the stream
probe of the box."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--variants", default="0,4")
    ap.add_argument("--extra-env", default="", help="VAR=a|b : a second axis, e.g. SOIL_CELLS_NT=0|1")
    a = ap.parse_args()
    from soillib_amd import _abi, soil
    _abi.check(_abi.lib().soil_set_device(0))
    param = bench.script_param(soil)
    run = bench._Single(a.size, a.size, param, 8, serial=False)
    ev = bench.Events(_abi, 6)
    for _ in range(2):
        run.step()
    run.sync()
    out = {}
    variants = [v for v in a.variants.split(",") if v]
    for rnd in range(a.rounds):
        for v in variants:
            os.environ["SOIL_CELLS_VARIANT"] = v
            run.step()                      # one untimed step on the new variant
            t = []
            for _ in range(a.steps):
                run.step(ev)
                t.append(ev.ms(2, 3))
            out.setdefault(v, []).append(sorted(t)[len(t) // 2])
    os.environ["SOIL_CELLS_VARIANT"] = "0"
    cells = a.size * a.size
    res = {"size": a.size, "median_ms_per_round": out,
           "TBps": {v: [round(112.0 * cells / (ms * 1e-3) / 1e12, 3) for ms in ts] for v, ts in out.items()}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
