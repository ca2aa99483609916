"""Host arithmetic of libsdpgpu.so that needs no GPU, through the C ABI: per-period layout, slabs and padding for any
world size, the arena sizes, the F1 window planner (reached through sdpgpu_keys_bytes), dependency footprints, halos
and state indexing -- on seeded random instances of every family and on the BASELINE grids at full size.
tests/test_sanitizers.py runs this file again on the AddressSanitizer + UBSan build of the library."""
import numpy as np
import pytest

import test_gpu_fuzz as tf
from stochastic_inventory_amd import workloads


def _engines(sia, w, world):
    out = []
    for r in range(world):
        d = w.desc()
        d.rank, d.world_size = r, world
        out.append(sia.SdpEngine(d, w.pmf, w.overhead()))
    return out


@pytest.mark.parametrize("family", [1, 2, 3, 4, 5, 6])
def test_random_instances_layout_slabs_arenas(sia, oracle, family):
    for seed in range(25):
        w = tf.make_instance(family, 700 + seed)
        world = 1 + seed % 5
        engs = _engines(sia, w, world)
        try:
            P = oracle.Problem(w.desc(), w.pmf, w.overhead())
            vbytes = 0
            for period in range(1, w.T + 1):
                S = engs[0].num_states(period)
                assert S == P.S[period - 1]
                slabs = [e.slab(period) for e in engs]
                pad = slabs[0][0]
                assert pad % world == 0 and S <= pad < S + world
                assert slabs[0][1] == 0 and slabs[-1][2] == S
                for a, b in zip(slabs[:-1], slabs[1:]):
                    assert a[0] == pad and a[2] == b[1] and a[1] <= a[2]
                    assert a[2] - a[1] in (pad // world, max(0, S - a[1]))
                vbytes += pad * 8
                # state index <-> tuple, both directions, on a sample
                x, cash, preq = P.state_arrays(period)
                preq2 = P.preq2_array(period)
                for idx in np.unique(np.linspace(0, S - 1, 7).astype(np.int64)):
                    assert engs[0].state_index(period, x[idx], cash[idx], preq[idx], preq2[idx]) == idx
            for e in engs:
                assert e.values_bytes() == vbytes
                kb = e.keys_bytes()
                assert kb == engs[0].keys_bytes()  # every rank takes the same decision (keys or fp64 rows travel)
                assert kb == 0 or kb == w.T * max(e.slab(p)[0] for p in range(1, w.T + 1)) * 8
                fp = [e.footprint(p) for p in range(1, w.T + 1)]
                assert fp == [engs[0].footprint(p) for p in range(1, w.T + 1)]
                for p, f in enumerate(fp, start=1):
                    if f is not None:
                        assert f[0] >= 0 and f[1] >= 0 and (p < w.T or f == (0, 0))
        finally:
            for e in engs:
                e.close()


@pytest.mark.parametrize("make,world", [(lambda: workloads.cfg2_clsp(), 1), (lambda: workloads.cfg2_clsp(), 8),
                                        (lambda: workloads.target_grid(), 1), (lambda: workloads.target_grid(), 8),
                                        (lambda: workloads.cfg5_scaled(S=100000000, T=3), 8),
                                        (lambda: workloads.cfg3_cash(), 4), (lambda: workloads.cfg3_tenths(), 8),
                                        (lambda: workloads.cfg4_leadtime(), 4), (lambda: workloads.cfg4_pipeline(), 4)],
                         ids=["cfg2x1", "cfg2x8", "targetx1", "targetx8", "cfg5x8", "cfg3x4", "cfg3tx8", "cfg4x4", "cfg4px4"])
def test_baseline_grids_full_size_geometry(sia, make, world):
    """The planner and the slab arithmetic at the sizes of BASELINE.json (no device memory is touched)."""
    w = make()
    engs = _engines(sia, w, world)
    try:
        T = w.T
        S = engs[0].num_states(1)
        assert sum(e.slab(1)[2] - e.slab(1)[1] for e in engs) == S
        kb = {e.keys_bytes() for e in engs}
        assert len(kb) == 1
        f = engs[0].footprint(1)
        if "f1" in w.name or "clsp" in w.name:
            D, A = len(w.pmf[0]), int(w.functor.maxOrderQuantity) + 1
            assert f == (D - 1, A - 1)  # demands 0..D-1, actions 0..A-1: state i reads V[i - (D-1) .. i + (A-1)]
            for e in engs[:2]:
                e.set_halo(3 * (A + D))
        else:
            assert f is None
        assert engs[0].values_bytes() == sum(engs[0].slab(p)[0] for p in range(1, T + 1)) * 8
        assert engs[-1].state_index(1, float("nan")) == -1
    finally:
        for e in engs:
            e.close()
