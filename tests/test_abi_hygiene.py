"""ABI 6 hygiene ("no C++ exception may cross the ABI").

(a) The two-product entry points (csrc/sdpgpu_sparse.hip) build host vectors; an exception there -- std::bad_alloc first of all
    -- comes back as SDPGPU_ERR_ALLOC / SDPGPU_ERR_INTERNAL with a message, never as std::terminate.  Forced here with the
    library's injection hook (SDPGPU_TEST_THROW, evaluated at the top of the guarded region, before any device call, so the
    check runs without a GPU); the real allocation of the memo read-out is capped in tests/test_gpu_multilead.py.
(b) sdpgpu_build_id: the binary says which sources it was built from, and it is THIS tree's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_reference.json")))

CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["SDP_ROOT"])
import stochastic_inventory_amd as sia
from stochastic_inventory_amd.multiitem import multilead_solve, multicash_solve, multixr_solve
k = json.load(open(os.path.join(os.environ["SDP_ROOT"], "tests", "golden", "kat_reference.json")))["kat1"]
KEYS = ("T", "q_bound", "price", "vari_cost", "sal_value", "ini_cash", "ini_i1", "ini_i2", "r0", "r1", "r2", "limit",
        "interest_free", "min_inventory", "max_inventory", "min_cash", "max_cash", "discount", "overhead", "values", "probs")
cash = dict(T=2, q_bound=4, price=[5, 10], vari_cost=[1, 2], sal_price=[0.5, 1], ini_cash=10, ini_i1=0, ini_i2=0, min_inventory=0,
            max_inventory=20, min_cash=0, max_cash=100, discount=1.0, pmf=[[[1, 1, 0.5], [2, 2, 0.5]]] * 2)
out = {}
for name, call in (("multilead", lambda: multilead_solve(**{n: k[n] for n in KEYS})), ("multicash", lambda: multicash_solve(**cash)),
                   ("multixr", lambda: multixr_solve(0.0, **cash))):
    try:
        call()
        out[name] = [0, ""]
    except sia.SdpgpuError as e:
        out[name] = [e.code, str(e)]
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("what,code,text", [("bad_alloc", 5, "host allocation failed"), ("runtime", 6, "injected (SDPGPU_TEST_THROW)"),
                                            ("other", 6, "unknown exception")])
def test_exceptions_in_the_two_product_entry_points_become_status_codes(what, code, text):
    env = dict(os.environ, SDP_ROOT=ROOT, SDPGPU_TEST_THROW=what)
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode == 0, f"the process died (terminate?): rc {out.returncode}\n{out.stderr[-1500:]}"
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    for name in ("multilead", "multicash", "multixr"):
        assert res[name][0] == code and text in res[name][1] and name in res[name][1], res


def test_build_id_is_the_digest_of_this_tree(sia):
    import __graft_entry__ as g
    g.build()
    from tools.kernel_sha import build_source_sha
    lib = sia._abi.load()
    bid = lib.sdpgpu_build_id().decode()
    assert len(bid) == 16 and int(bid, 16) >= 0 and bid == build_source_sha(ROOT)
    assert g.check_build_id(lib) == bid
    # the identity can be read out of the file without loading it (what build.py's up-to-date check does)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "stochastic-inventory_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.baked_build_id() == bid and b.up_to_date()


def test_a_stale_library_fails_the_check(sia, monkeypatch):
    import __graft_entry__ as g
    import tools.kernel_sha as ks
    monkeypatch.setattr(ks, "build_source_sha", lambda root: "0" * 16)
    with pytest.raises(AssertionError, match="was not built from"):
        g.check_build_id(sia._abi.load())
