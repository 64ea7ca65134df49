#!/usr/bin/env python
"""Why does k_erode_cells_fused take 1.55 ms inside the bench step and 1.25 ms in a loop by itself?
Times the same launch on the same planes in several contexts, one process, one box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    from soillib_amd import _abi, soil, silt
    lib = _abi.lib()
    _abi.check(lib.soil_set_device(0))
    param = bench.script_param(soil)
    run = bench._Single(size, size, param, 8, serial=False)
    m = run.model
    ev = bench.Events(_abi, 6)
    for _ in range(3):
        run.step()
    run.sync()
    res = {}

    def med(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2]

    # (a) inside the step
    t = []
    for _ in range(6):
        run.step(ev)
        t.append(ev.ms(2, 3))
    res["in_step"] = med(t)
    # (b) back to back, same planes (events around each launch)
    t = []
    for _ in range(12):
        ev.record(0)
        m.cells_fused()
        ev.record(1)
        run.sync()
        t.append(ev.ms(0, 1))
    res["back_to_back"] = med(t[2:])
    # (b2) 10 launches between one pair of events
    ev.record(0)
    for _ in range(10):
        m.cells_fused()
    ev.record(1)
    run.sync()
    res["ten_in_a_row_avg"] = ev.ms(0, 1) / 10
    # (c) particle phase, host sleeps 20 ms with the GPU idle, then cells
    t = []
    for _ in range(5):
        m.seed_step()
        m.particles_pair()
        run.sync()
        time.sleep(0.02)
        ev.record(0)
        m.cells_fused()
        ev.record(1)
        run.sync()
        t.append(ev.ms(0, 1))
        m.swap_layers()
        m.step_index += 1
    res["after_particles_and_idle_20ms"] = med(t)
    # (d) particle phase, then cells twice: is the second one fast?
    t1, t2 = [], []
    for _ in range(5):
        m.seed_step()
        m.particles_pair()
        ev.record(0)
        m.cells_fused()
        ev.record(1)
        m.cells_fused()
        ev.record(2)
        run.sync()
        t1.append(ev.ms(0, 1))
        t2.append(ev.ms(1, 2))
        m.swap_layers()
        m.step_index += 1
    res["after_particles_first"] = med(t1)
    res["after_particles_second"] = med(t2)
    # (e) sequential particle launches on the caller's stream (no forked streams), then cells
    t = []
    for _ in range(5):
        m.seed_step()
        m.particles_fluvial()
        m.particles_debris()
        ev.record(0)
        m.cells_fused()
        ev.record(1)
        run.sync()
        t.append(ev.ms(0, 1))
        m.swap_layers()
        m.step_index += 1
    res["after_sequential_particles"] = med(t)
    # (f) only the debris launch before (short), then cells
    t = []
    for _ in range(5):
        m.seed_step()
        m.particles_debris()
        ev.record(0)
        m.cells_fused()
        ev.record(1)
        run.sync()
        t.append(ev.ms(0, 1))
    res["after_debris_only"] = med(t)
    # (g) a plain stream over two other planes before the cells
    a = silt.tensor(silt.float32, silt.shape(8192, 8192), silt.gpu)
    b = silt.tensor(silt.float32, silt.shape(8192, 8192), silt.gpu)
    silt.set(a, 0.0)
    silt.set(b, 1.0)
    t = []
    for _ in range(6):
        for _ in range(4):
            silt.add(a, b)
        ev.record(0)
        m.cells_fused()
        ev.record(1)
        run.sync()
        t.append(ev.ms(0, 1))
    res["after_stream_kernels"] = med(t)
    print(json.dumps({k: round(v, 4) for k, v in res.items()}))


if __name__ == "__main__":
    main()
