"""The cash quantiser's rounding on the device (csrc/sdp_cash.hpp: jround_rtn*, two fp64 additions under round-toward-minus-
infinity inside one asm block) against Math.round's definition -- floor(x + 1/2) in exact arithmetic, CashConstraint.java:131
`Math.round(cash * 10)` -- on adversarial values (the predecessor of 0.5, every tie in a range and its neighbours, the ends of the
admitted range) and four million random ones; and the wave's rounding mode is round-to-nearest again behind the block.
tests/hip/round_rtn_check.hip is compiled here with hipcc and run on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rtn_rounding_is_math_round(tmp_path):
    exe = tmp_path / "round_rtn_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                    "-Wno-unused-function", "-o", str(exe), os.path.join(ROOT, "tests", "hip", "round_rtn_check.hip")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 mismatches, 0 wide-form disagreements, 0 with a wrong rounding mode" in out.stdout, out.stdout
