"""docs/KNOBS.md is the one table of the SOIL_* environment variables; it is generated from the getenv /
os.environ call sites and must match them (the knobs were scattered over 160 KB of prose)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_knob_table_matches_the_call_sites():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_knobs.py"), "--check"], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    text = open(os.path.join(ROOT, "docs", "KNOBS.md")).read()
    for name in ("SOIL_RCCL_TIMEOUT_S", "SOIL_TILED_STEPS", "SOIL_RAKE_LIST_FROM", "SOIL_STEP_PAIR"):
        assert "`%s`" % name in text
