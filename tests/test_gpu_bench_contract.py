"""bench.py's one-line JSON contract at N = 1 (the line the driver parses): every key, its type and the relations
between them, on a short run of the default workload and of one other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--cpu-seconds", "2", *extra], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--workload", "cfg3", "--periods", "3"]], ids=["default_cfg2", "cfg3"])
def test_bench_line(extra):
    r = _run(*extra)
    assert r["metric"] == "(state,action,demand) cell evals/sec" and r["unit"] == "cells/s"
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic"
    assert isinstance(r["config"]["workload"], str) and "model" not in r["config"]
    assert r["value"] > 1e11 and r["ms_per_step"] > 0
    # value = cells of one sweep / time of one sweep
    assert abs(r["value"] - r["config"]["cells_per_step"] / (r["ms_per_step"] * 1e-3)) <= 0.02 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["unit"] == "cells/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert isinstance(cb["sample"], str) and cb["sample"]
