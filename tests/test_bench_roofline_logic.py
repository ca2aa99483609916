"""bench.py's roofline block without a GPU: the binding unit is the one closest to its peak, every fraction is a fraction,
and a figure outside (0, 1] is withheld with its reason instead of printed (a `frac` of 6.75)."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stats(fp64_ops=0.0, lds=0.0, l1=0.0, r=0, s=0):
    return types.SimpleNamespace(fp64_ops_executed=fp64_ops, lds_bytes=lds, l1_bytes=l1, window_r=r, window_s=s)


def _workload(name):
    return types.SimpleNamespace(name=name)


def test_fp64_issue_binds_the_window_kernel():
    cells = 6e11
    st = _stats(fp64_ops=cells * 3.448, lds=cells * 1.0, r=4, s=4)
    rf = bench.roofline_block(_workload("no_profile_for_this_name"), st, 6, cells, 6e6, 55.9, [9.3] * 6)
    assert rf["bound"] == "fp64-valu" and abs(rf["frac"] - cells * 3.448 / 55.9e-3 / bench.VALU_PEAK_LANE_OPS) < 1e-12
    assert 0.9 < rf["frac"] < 1.0 and rf["units"]["lds"]["frac"] < rf["frac"]
    assert rf["hbm"] is None and rf["traffic"] is None  # no counter summary for that name
    assert rf["algorithmic"]["frac_of_hbm_peak"] > 1.0  # the byte MODEL may exceed the HBM peak; it is labelled as such


def test_lds_binds_the_diagonal_cash_kernel():
    cells = 2.6e11
    st = _stats(fp64_ops=cells * 3.0, lds=cells * 9.5, l1=cells * 1.5)
    rf = bench.roofline_block(_workload("no_profile_for_this_name"), st, 6, cells, 6e6, 38.9, [6.5] * 6)
    assert rf["bound"] == "lds" and rf["unit"] == "TB/s" and abs(rf["peak"] - 78.6432) < 1e-9
    assert set(rf["units"]) == {"lds", "vector-l1", "fp64-valu"}
    assert all(0.0 < u["frac"] <= rf["frac"] for u in rf["units"].values())


def test_a_fraction_above_one_is_withheld(capsys):
    cells = 1e10
    st = _stats(fp64_ops=cells * 50.0)  # an op model that cannot be true at this speed
    rf = bench.roofline_block(_workload("no_profile_for_this_name"), st, 2, cells, 1e4, 0.01, [0.005] * 2)
    assert rf["frac"] is None and rf["achieved"] is None and "withheld" in rf["error"]
    assert "withheld" in capsys.readouterr().err
    assert rf["avg_launch_ms"] == 0.005 and len(rf["per_launch_ms_events"]) == 2  # the measured times stay


def test_every_bench_workload_has_a_kernel_source_digest():
    """tools/kernel_sha.py maps every --workload of bench.py to the files its kernels are built from (a counter summary in
    profiles/ is used only when that digest matches): each family to its own files, all of them present."""
    import os
    from tools.kernel_sha import kernel_files, kernel_source_sha, build_source_sha
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "stochastic-inventory_amd", "csrc")
    names = {"target_f1_1000000x500x200x6": "sdp_window.hpp", "cfg2_clsp_x": "sdp_window.hpp", "cfg5_f1_x": "sdp_window.hpp",
             "cfg3_cash_x": "sdp_cash.hpp", "cfg3t_cash_tenths_x": "sdp_cash.hpp", "f5_spl_x": "sdp_cash.hpp",
             "separable_f5_spl_x": "sdp_cash.hpp", "separable_target_f1_x": "sdp_window.hpp", "cfg4_leadtime_x": "sdp_window.hpp",
             "staff_testing0_x": "sdp_staff.hpp", "multilead_kat2_T3_Q50": "sdpgpu_sparse.hip", "custom_clsp_x": "sdp_custom_src.hpp",
             "custom_clsp_level_x": "sdp_window.hpp"}
    for name, must in names.items():
        files = kernel_files(name)
        assert must in files, (name, files)
        assert all(os.path.exists(os.path.join(csrc, f)) for f in files)
        assert len(kernel_source_sha(root, name)) == 16
    assert "sdp_custom_src.hpp" in kernel_files("custom_clsp_level_x")
    assert len(build_source_sha(root)) == 16


def test_a_sweep_of_different_kernels_is_priced_against_the_counted_kernels_own_duration(monkeypatch):
    """The workforce family's periods run different kernel forms on staff ranges of 1 to 7001 numbers: the mean of the sweep's
    launches is not the counted kernel's duration.  The counted instructions are then divided by that kernel's own duration in
    the profile (and the note says so); a sweep of like launches keeps the HIP-event time."""
    insts = 3.1e8  # wave instructions of the counted kernel per launch
    pmc = {"_file": "profiles/x.json", "dominant_kernel": "k<4, 4, true>", "valu_insts_per_launch": insts,
           "kernels": {"k<4, 4, true>": {"rocprof": {"calls": 35, "avg_us": 663.0}}}}
    monkeypatch.setattr(bench, "load_pmc", lambda name: dict(pmc))
    per_launch = [0.66, 0.87, 0.83, 0.72, 0.46, 0.44, 0.29, 0.07]
    rf = bench.roofline_block(_workload("staff_x"), _stats(), 8, 2.7e10, 3e4, sum(per_launch), per_launch)
    assert rf["bound"] == "valu-issue" and abs(rf["dominant_launch_ms"] - 0.663) < 1e-9 and "own rocprofv3 duration" in rf["note"]
    assert abs(rf["frac"] - insts * 64 / 0.663e-3 / bench.VALU_PEAK_LANE_OPS) < 1e-12 and 0.7 < rf["frac"] < 0.8
    like = [0.74, 41.6, 41.7, 41.7]
    pmc["kernels"]["k<4, 4, true>"]["rocprof"]["avg_us"] = 41990.0
    pmc["valu_insts_per_launch"] = 2.0e10
    rf = bench.roofline_block(_workload("f5_x"), _stats(), 4, 3.56e11, 1e8, sum(like), like)
    assert abs(rf["dominant_launch_ms"] - sum(like[1:]) / 3) < 1e-9 and "HIP-event launch time" in rf["note"]
