"""CPU-side checks of the boundary: the shared library loads, exports every symbol the header
declares, and the host logic that needs no GPU (validation, layout, error reporting) behaves."""
import ctypes as C
import os
import re

import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(sia):
    import __graft_entry__ as g
    g.build()
    return sia._abi.load()


def test_every_declared_symbol_is_exported(sia, lib):
    header = open(os.path.join(ROOT, "include", "sdpgpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sdpgpu_[a-z0-9_]+)\s*\(", header))
    assert declared == set(sia._abi.EXPORTS), declared ^ set(sia._abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)


def test_struct_layout_matches_the_header(sia, lib):
    d = sia.SdpgpuDesc()
    lib.sdpgpu_desc_init(C.byref(d))
    ref = sia.desc_defaults()
    assert bytes(d) == bytes(ref)
    # sizeof / offsetof as the C compiler sees the header
    import subprocess, tempfile
    fields = [f[0] for f in sia.SdpgpuDesc._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "sdpgpu.h"\nint main(void){printf("%zu", sizeof(sdpgpu_desc));' + \
        "".join(f'printf(" %zu", offsetof(sdpgpu_desc, {f}));' for f in fields) + \
        'printf(" %zu", sizeof(sdpgpu_stats)); return 0;}'
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "a.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "a"), os.path.join(td, "a.c")], check=True)
        out = subprocess.run([os.path.join(td, "a")], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == C.sizeof(sia.SdpgpuDesc)
    assert [int(v) for v in out[1:-1]] == [getattr(sia.SdpgpuDesc, f).offset for f in fields]
    assert int(out[-1]) == C.sizeof(sia.SdpgpuStats)
    assert d.abi_version == 6 and d.discount_factor == 1.0 and d.cash_round_div == 10.0 and d.world_size == 1


def test_two_product_struct_layouts_match_the_header(sia):
    """sdpgpu_multilead / sdpgpu_multicash / sdpgpu_multi_table: size and every field offset as gcc sees the header."""
    import subprocess, tempfile
    from stochastic_inventory_amd import _abi
    pairs = [("sdpgpu_multilead", _abi.SdpgpuMultilead), ("sdpgpu_multicash", _abi.SdpgpuMulticash),
             ("sdpgpu_multi_table", _abi.SdpgpuMultiTable)]
    body = ""
    for cname, cls in pairs:
        body += f'printf("%zu", sizeof({cname}));' + "".join(f'printf(" %zu", offsetof({cname}, {f[0]}));' for f in cls._fields_) + 'printf("\\n");'
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "sdpgpu.h"\nint main(void){' + body + 'return 0;}'
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "a.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "a"), os.path.join(td, "a.c")], check=True)
        lines = subprocess.run([os.path.join(td, "a")], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for (cname, cls), line in zip(pairs, lines):
        out = [int(v) for v in line.split()]
        assert out[0] == C.sizeof(cls), cname
        assert out[1:] == [getattr(cls, f[0]).offset for f in cls._fields_], cname


def test_create_rejects_bad_descriptors(sia, lib):
    d = sia.desc_defaults()
    d.family = 9
    h = C.c_void_p()
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 1
    assert b"family" in lib.sdpgpu_last_error(None)
    d = sia.desc_defaults()
    d.step = 3.0
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 4  # unsupported, says why
    assert b"step" in lib.sdpgpu_last_error(None)
    d = sia.desc_defaults()
    d.family, d.direction = sia.FAMILY_LEADTIME, 1
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 1
    assert b"MIN only" in lib.sdpgpu_last_error(None)


@pytest.mark.parametrize("make", cases.ALL, ids=lambda f: f.__name__)
def test_layout_agrees_with_the_oracle(sia, oracle, make):
    """Two independent layout implementations (product C++ vs oracle C) give the same grids."""
    w = make()
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
    for period in range(1, w.T + 1):
        g = P.grids[period - 1]
        assert eng.grid(period) == (g.x_lo, g.nx, g.nc, g.nq)
        assert eng.num_states(period) == P.S[period - 1]
        x, cash, preq = P.state_arrays(period)
        preq2 = P.preq2_array(period)
        assert eng.grid2(period) == (g.x_lo, g.nx, g.nc, g.nq1, g.nq // g.nq1)
        for idx in (0, len(x) // 3, len(x) - 1):
            assert eng.state_index(period, x[idx], cash[idx], preq[idx], preq2[idx]) == idx
            if g.nc > 1 and w.desc().cash_formula != 2:
                assert eng.cash_value(idx % g.nc) == cash[idx]
            elif g.nc > 1:  # the (x, R) state: the tuple's second entry is R = grid cash + variCost * x
                assert eng.cash_value(idx % g.nc) + w.functor.variCost * x[idx] == cash[idx]
    assert eng.state_index(1, 0.5, 0.0, 0.0) == -1
    eng.close()


def test_slabs_partition_the_grid(sia):
    w = cases.f3_tenths()
    seen = []
    for rank in range(3):
        d = w.desc()
        d.rank, d.world_size = rank, 3
        eng = sia.SdpEngine(d, w.pmf, w.overhead())
        pad, lo, hi = eng.slab(1)
        assert pad % 3 == 0 and pad >= eng.num_states(1)
        seen.append((lo, hi))
        eng.close()
    assert seen[0][0] == 0 and seen[-1][1] == 11 * 231
    assert all(seen[i][1] == seen[i + 1][0] for i in range(2))


def test_create_rejects_the_new_descriptor_fields_misused(sia, lib):
    h = C.c_void_p()
    d = sia.desc_defaults()
    d.lead_time = 2  # a two-stage pipeline exists for the LEADTIME family only
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 4 and b"LEADTIME" in lib.sdpgpu_last_error(None)
    d.family, d.clamp_inventory = sia.FAMILY_LEADTIME, 0
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 4 and b"clamp" in lib.sdpgpu_last_error(None)
    d.lead_time = 3
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 1
    d = sia.desc_defaults()
    d.family, d.direction = sia.FAMILY_SURVIVAL, 0  # getSurvProb maximises
    d.cash_round_mult = d.cash_round_div = 1.0
    d.max_cash = 10.0
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 1 and b"maximises" in lib.sdpgpu_last_error(None)
    d.direction = 1
    assert lib.sdpgpu_create(C.byref(d), C.byref(h)) == 0
    lib.sdpgpu_destroy(h)
    prm = (C.c_double * 300)()
    assert lib.sdpgpu_create_custom(C.byref(d), b"", prm, 300, C.byref(h)) == 1  # more than 256 parameters
    assert lib.sdpgpu_create_custom(C.byref(d), None, prm, 1, C.byref(h)) == 1


def test_errors_are_reported_not_thrown(sia):
    w = cases.f1_small()
    with pytest.raises(ValueError):
        sia.SdpEngine(w.desc(), w.pmf[:-1])
    bad = [t.copy() for t in w.pmf]
    bad[0][:, 0] = bad[0][::-1, 0]  # descending demands
    with pytest.raises(sia.SdpgpuError) as e:
        sia.SdpEngine(w.desc(), bad)
    assert "ascending" in str(e.value)
    eng = sia.SdpEngine(w.desc(), w.pmf)
    with pytest.raises(sia.SdpgpuError):
        eng.values(1)  # nothing solved yet
    eng.close()


def test_compute_fails_loudly_without_a_gpu(sia):
    """No HIP device in the authoring container: the product must refuse, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    w = cases.f1_small()
    eng = sia.SdpEngine(w.desc(), w.pmf)
    with pytest.raises(sia.SdpgpuError) as e:
        eng.solve()
    assert e.value.code == 3
    eng.close()


def test_host_mirror_keeps_the_reference_api(sia):
    assert callable(sia.RiskRecursion.getSurvProb) and callable(sia.RiskSimulation.simulateLostSale)
    r = sia.RiskState(1, 0.0, 5.0, True)
    assert r.getBankruptBefore() is False and r == sia.RiskState(1, 0.0, 5.0, False)  # RiskState.java:17
    for cls in (sia.Recursion, sia.CashRecursion, sia.LeadtimeRecursion, sia.CashLeadtimeRecursion):
        for name in ("getExpectedValue", "getAction", "getCacheActions", "getOptTable",
                     "getStateTransitionFunction", "getImmediateValueFunction", "setTreeMapCacheAction"):
            assert callable(getattr(cls, name))
    s = sia.CashState(2, 3.0, 4.5)
    assert (s.getPeriod(), s.getIniInventory(), s.getIniCash()) == (2, 3.0, 4.5)
    assert s == sia.CashState(2, 3.0, 4.5) and s != sia.CashState(2, 3.0, 4.6)
    assert sia.State(1, 0.0) != sia.LeadtimeState(1, 0.0, 0.0)
    with pytest.raises(TypeError):
        sia.Recursion(sia.OptDirection.MIN, cases.f1_small().pmf)  # functor descriptor is mandatory


def test_java_round_host(sia):
    assert sia.java_round(-2.5) == -2 and sia.java_round(2.5) == 3 and sia.java_round(0.49999999999999994) == 0


@pytest.mark.parametrize("make", [cases.f1_small, cases.f3_tenths, cases.f5_cash_leadtime, cases.f2_pipeline],
                         ids=lambda m: m.__name__)
def test_state_index_rejects_non_finite_and_huge_inputs(sia, make):
    """NaN / infinite / astronomically large coordinates are 'not a grid point' (-1), never an index: the range checks
    come before the double -> int64 casts (which are undefined for such inputs)."""
    w = make()
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        x_lo, nx, nc, nq1, nq2 = eng.grid2(1)
        ok = eng.state_index(1, x_lo, eng.cash_value(0) if nc > 1 else 0.0, 0.0, 0.0)
        assert ok == 0
        for bad in (float("nan"), float("inf"), -float("inf"), 1e300, -1e300, 2.0 ** 63, -(2.0 ** 63)):
            assert eng.state_index(1, bad, eng.cash_value(0) if nc > 1 else 0.0, 0.0, 0.0) == -1
            if nc > 1:
                assert eng.state_index(1, x_lo, bad, 0.0, 0.0) == -1
            if nq1 > 1:
                assert eng.state_index(1, x_lo, eng.cash_value(0) if nc > 1 else 0.0, bad, 0.0) == -1
            if nq2 > 1:
                assert eng.state_index(1, x_lo, 0.0, 0.0, bad) == -1
