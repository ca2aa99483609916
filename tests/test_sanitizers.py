"""CPU sanitizer runs (GPU AddressSanitizer / XNACK are not available on the pool, so these cover host code only):

1. the ORACLE (oracle/sdpref.c, oracle/staffref.c) built with gcc -fsanitize=address,undefined and driven by its own
   test-suite in a child process with libasan preloaded;
2. the HOST half of libsdpgpu.so -- descriptor validation, per-period layout, slab / halo arithmetic, the F1 window
   planner, footprints, state indexing, error paths -- built with ROCm clang's -fsanitize=address,undefined
   (-fno-gpu-sanitize) and driven through the C ABI by tests/test_abi.py's geometry and validation tests.

A run fails on any sanitizer report (UBSan is built non-recoverable; ASan aborts the child)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_child(env_extra, files, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    # leak checking is off: CPython itself "leaks" by LeakSanitizer's standards
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:verify_asan_link_order=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *files],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    out = r.stdout[-4000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out
    assert r.returncode == 0, out
    assert " passed" in r.stdout
    return r.stdout


def test_oracle_suite_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so here")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libsdpref_asan.so", "libstaffref_asan.so"], check=True)
    files = ["tests/test_oracle_kat.py", "tests/test_oracle_selfconsistency.py", "tests/test_staff_oracle.py",
             "tests/test_golden.py", "tests/test_multicash_oracle.py", "tests/test_xr_oracle.py"]
    files = [f for f in files if os.path.exists(os.path.join(ROOT, f))]
    out = _run_child({"SDPREF_SANITIZE": "1", "LD_PRELOAD": os.path.realpath(libasan)}, files, timeout=900)
    assert "failed" not in out


def test_host_half_of_libsdpgpu_under_asan_ubsan():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_sdp_build", os.path.join(ROOT, "stochastic-inventory_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    lib = b.build_host_asan()
    out = _run_child({"SDPGPU_LIB": lib, "LD_PRELOAD": b.asan_runtime()},
                     ["tests/test_abi.py", "tests/test_planner_geometry.py"], timeout=900)
    assert "failed" not in out
