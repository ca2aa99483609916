"""configs[4] at full width on one GPU (1e8 states x 500 actions x 200 demands, 2 periods = 2e13 cells):
the oracle cannot sweep that, so 3000 sampled states per period are compared with the oracle fed the GPU's
own V_{t+1} -- bit for bit -- plus the clamp edges of the grid."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_1e8_state_grid_sampled_parity():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_grid_check.py"), "100000000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("bit-identical") == 2


def test_1e7_state_pipeline_grid_sampled_parity():
    """configs[3] at full size: (x, q1, q2) = 250 x 200 x 200 states x 200 actions x 100 demands, 3 periods."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pipeline_grid_check.py"), "3"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("bit-identical") == 3


@pytest.mark.parametrize("name,periods", [("cfg3", 4), ("cfg4", 4)])
def test_full_size_config_sampled_parity(name, periods):
    """configs[2] (1e6 cash states x <=300 actions x 150 demands, 6 periods) and configs[3] in the reference's
    (period, x, preQ) shape (50 periods x 1000 x 200) at full size: sampled states of four periods against the
    oracle fed the GPU's own V_{t+1}."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sampled_grid_check.py"), name, "1500"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("bit-identical") == periods
