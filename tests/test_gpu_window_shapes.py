"""The window kernels' launch shapes (csrc/window.hpp, SOIL_WIN_SHAPE) are picked by grid size: the
parity grids are small and would only ever see the `small` choice.  The variable is read once per
process, so each shape runs the flow-map / stencil / seam / accumulate parity tests in a process of
its own: band walk (0), rows through LDS (1), flat (2), bands of 16 / 8 rows (3 / 6), blocks of four /
two rows (4 / 5), the work-group's waves stacked over one strip (7 / 8)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARITY = ("test_flow_maps_bit_exact or test_stencils_bit_exact or test_window_kernels_across_seams "
          "or test_steepest_ties_and_near_ties or test_accumulate_bit_exact")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_parity_under_every_window_shape(hip, shape):
    env = dict(os.environ, SOIL_WIN_SHAPE=str(shape))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q",
                        "-m", "gpu", "-p", "no:cacheprovider", "-k", PARITY],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "SOIL_WIN_SHAPE=%d:\n%s\n%s" % (shape, r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
