"""`python bench.py --config c1|c2|c3` print ONE JSON line each in the driver's format, with `roofline`
and `cpu_baseline` (BASELINE.json configs[0..2]; c4 = the default line, tests/test_bench_spawn.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config")


def _line(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SOIL_BENCH_GRID")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    for k in KEYS:
        assert k in out, k
    assert "workload" in out["config"] and "model" not in out["config"]
    return out


def test_config1_line_runs_on_the_host():
    """configs[0] is a CPU path in the reference and here: the line exists without a GPU (its roofline
    block — the device twin of the operator — is then null)."""
    out = _line("--config", "c1", "--steps", "10", "--warmup", "1")
    assert out["value"] > 0 and out["config"]["grid"] == [256, 256] and out["unit"] == "Mcells/s"
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["equals_product"] is True


@pytest.mark.gpu
def test_config_lines_on_gpu(hip):
    c1 = _line("--config", "c1", "--steps", "20")
    r = c1["roofline"]
    assert r["bound"] == "hbm" and r["algorithmic_bytes_per_cell"] == 16 and r["device_equals_host_bit_for_bit"] is True
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c2 = _line("--config", "c2", "--steps", "20", "--warmup", "3", "--cpu-size", "256")
    assert c2["config"]["grid"] == [1024, 1024] and c2["config"]["baseline_config"] == "configs[1]"
    assert c2["config"]["particle_arithmetic"].startswith("exact") and "fast_arithmetic" in c2
    assert c2["roofline"]["kernel"] == "k_erode_cells_fused" and c2["cpu_baseline"]["value"] > 0
    c3 = _line("--config", "c3", "--size", "512", "--steps", "16", "--warmup", "2")
    assert c3["config"]["K"] == 16 and c3["config"]["grid"] == [512, 512] and c3["value"] > 0
    r = c3["roofline"]
    assert r["bound"] == "hbm" and r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_cell"] > 0
    assert c3["config"]["mean_upstream_area_min_max"][0] >= 1.0        # every cell drains at least itself
    assert c3["cpu_baseline"]["cores"] == 1
