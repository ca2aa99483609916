"""bench.py's one-line JSON contract at N = 1 (the line the driver parses): every key, its type and the relations
between them, on a short run of the default workload (reduced horizon) and of two others."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--cpu-seconds", "2", *extra], capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def _check_roofline(rf):
    # the unit that binds the kernel: fp64 issue (lane-op/s), or an on-chip data path (LDS 128 B/clk/CU, vector L1 64 B/clk/CU)
    peaks = {"fp64-valu": ("T lane-op/s", 39.3216), "valu-issue": ("T lane-op/s", 39.3216), "lds": ("TB/s", 78.6432),
             "vector-l1": ("TB/s", 39.3216)}
    assert rf["bound"] in peaks and rf["unit"] == peaks[rf["bound"]][0]
    assert abs(rf["peak"] - peaks[rf["bound"]][1]) < 1e-6
    for name, u in rf.get("units", {}).items():
        assert 0.0 < u["frac"] <= 1.0 and abs(u["frac"] - u["achieved"] / u["peak"]) < 1e-9, name
        assert rf["frac"] is None or u["frac"] <= rf["frac"] + 1e-12, "`bound` names the unit closest to its peak"
    if rf["frac"] is not None:
        assert 0.0 < rf["frac"] <= 1.0, "a roofline fraction is a fraction"
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert rf["hbm"] is None or 0.0 < rf["hbm"]["frac"] <= 1.0
    assert len(rf["per_launch_ms_events"]) == rf["launches_per_sweep"]
    assert rf["algorithmic"]["bytes_per_launch"] > 0


@pytest.mark.parametrize("extra", [["--periods", "3", "--no-secondary"], ["--workload", "cfg2"],
                                   ["--workload", "cfg3", "--periods", "3"]], ids=["default_target_T3", "cfg2", "cfg3"])
def test_bench_line(extra):
    r = _run(*extra)
    assert r["metric"] == "(state,action,demand) cell evals/sec" and r["unit"] == "cells/s"
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "strong" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic"
    assert isinstance(r["config"]["workload"], str) and "model" not in r["config"]
    if "default" in " ".join(extra) or not [e for e in extra if e == "--workload"]:
        assert r["config"]["workload"].startswith("target_f1_1000000x500x200")
    assert r["value"] > 1e11 and r["ms_per_step"] > 0
    # value = cells of one sweep / time of one sweep
    assert abs(r["value"] - r["config"]["cells_per_step"] / (r["ms_per_step"] * 1e-3)) <= 0.02 * r["value"]
    assert r["parity_gate"]["status"] == "ok" and r["parity_gate"]["states_checked"] > 100
    _check_roofline(r["roofline"])
    # the per-launch event times of one sweep add up to (about) one sweep
    assert sum(r["roofline"]["per_launch_ms_events"]) <= 1.15 * r["ms_per_step"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["unit"] == "cells/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert isinstance(cb["sample"], str) and cb["sample"]


def test_headline_is_the_target_grid_with_secondaries():
    """The default line: the 1e6 x 500 x 200 grid at T = 6 with the other BASELINE configs as gated secondary entries."""
    r = _run("--steps", "2")
    assert r["config"]["workload"] == "target_f1_1000000x500x200x6" and r["config"]["cells_per_step"] == 6 * 10 ** 11
    assert r["roofline"]["bound"] == "fp64-valu" and r["roofline"]["frac"] > 0.5
    names = [s["workload"] for s in r["secondary"]]
    assert any(n.startswith("cfg2_clsp_10000x200x100x52") for n in names) and any(n.startswith("cfg3_cash") for n in names)
    assert any(n.startswith("cfg3t_cash_tenths") for n in names) and any(n.startswith("cfg4_leadtime") for n in names)
    # round 4: configs[4] at full width, the SURVEY 8(f)-3 families at the sizes of the reference's slowest drivers, the 8(f)-4 mode
    for prefix in ("cfg5_f1_100000000x500x200x3", "f5_spl_", "staff_testing0", "custom_clsp_10000x200x100x52", "multilead_kat2",
                   "separable_target_f1_1000000x500x200x6"):
        assert any(n.startswith(prefix) for n in names), (prefix, names)
    for s in r["secondary"]:
        assert s["parity_gate"]["status"] == "ok", s["workload"]
        assert s["value"] > 0 and s["ms_per_step"] > 0
        if s["workload"].startswith("multilead"):  # (a solve, not a sweep of launches: its own roofline block)
            rf = s["roofline"]
            assert rf["bound"] == "valu-issue" and (rf["frac"] is None or 0.0 < rf["frac"] <= 1.0)
            assert s["parity_gate"]["final_cash"] == -76.56 and s["parity_gate"]["first_order"] == [30, 15]
        else:
            _check_roofline(s["roofline"])
    sep = [s for s in r["secondary"] if s["workload"].startswith("separable_target")][0]
    assert sep["speedup_over_brute_force"] > 3 and abs(sep["ms_per_sweep"] - sep["ms_per_step"]) < 1e-9
    assert sep["parity_gate"]["worst_relative_difference"] <= 1e-9
    assert r["build"]["match"] is True and len(r["build"]["build_id"]) == 16


def test_a_failing_secondary_entry_does_not_take_the_headline_down(monkeypatch):
    """A secondary workload whose gate fails (injected) is recorded in its place -- no value, the reason -- and the line, whose
    headline has passed its own gate, is still printed with the other entries intact."""
    monkeypatch.setenv("SDP_BENCH_ONLY_SECONDARY", "cfg2,staff,custom_clsp_level")
    monkeypatch.setenv("SDP_BENCH_TEST_FAIL_SECONDARY", "staff")
    r = _run("--steps", "2")
    assert r["parity_gate"]["status"] == "ok" and r["value"] > 1e12
    by = {s["workload"].split("_")[0]: s for s in r["secondary"]}
    assert set(by) == {"cfg2", "staff", "custom"}
    assert by["staff"]["value"] is None and by["staff"]["parity_gate"]["status"] == "FAILED" and "injected" in by["staff"]["error"]
    assert by["cfg2"]["parity_gate"]["status"] == "ok" and by["custom"]["parity_gate"]["status"] == "ok"
