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


def _plan(sia, w, monkeypatch, env=None, store_all=1, world=1):
    for k in ("SDPGPU_WIN_R", "SDPGPU_WIN_S", "SDPGPU_WIN_NCH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    d = w.desc()
    d.store_all_values = store_all
    d.world_size = world
    with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
        return eng.plan(1)


def test_window_plans_use_the_whole_lds_of_a_compute_unit(sia, monkeypatch):
    """gfx950 has 160 KiB of LDS per compute unit: the one-task-per-tile plan of the 500-action x 200-demand grid on the
    (4, 8) block is 78.4 KiB per workgroup and two of them are resident (sdpgpu_plan_period, host arithmetic only)."""
    pl = _plan(sia, workloads.target_grid(T=2), monkeypatch, {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "1"})
    assert (pl.kernel, pl.r, pl.s, pl.chunks, pl.chunk_blocks) == (2, 4, 8, 1, 125)
    assert pl.tiles == pl.tasks == 1954 and pl.lds_bytes == 80288 and pl.workgroups_per_cu == 2
    # ping-pong tables (configs[4] on one GPU) cannot keep chunk rows: the planner now takes the (4, 8) block there as well
    pl = _plan(sia, workloads.cfg5_scaled(S=12500000, T=3), monkeypatch, store_all=0)
    assert (pl.r, pl.s, pl.chunks) == (4, 8, 1) and 65536 < pl.lds_bytes <= 81920
    # with resident tables the 1e6-state grid keeps several chunks per tile (1954 one-chunk tasks would leave 5 % of the
    # 2048 wave slots idle; measured 56.4 against 53.3 ms per sweep) -- same on every rank of eight
    pl1 = _plan(sia, workloads.target_grid(), monkeypatch)
    pl8 = _plan(sia, workloads.target_grid(), monkeypatch, world=8)
    assert (pl1.r, pl1.s) == (4, 8) and pl1.chunks > 1 and pl1.lds_bytes <= 65536
    assert pl8.chunks > 1 and pl8.tiles == (125000 + 511) // 512


@pytest.mark.parametrize("env,needle", [
    ({"SDPGPU_WIN_R": "7"}, "no instantiation"),
    ({"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "1"}, "of LDS per workgroup"),
])
def test_infeasible_forced_window_plans_are_refused_with_a_reason(sia, monkeypatch, env, needle):
    """A forced plan that cannot run is an SDPGPU_ERR_ARG with a message from the planner, not a launch failure:
    an R the kernel is not instantiated for; 3000 actions x 400 demand steps in ONE chunk (251 KiB of windows)."""
    w = workloads.cfg5_scaled(S=100000, T=2, A=3000, D=400)
    with pytest.raises(sia.SdpgpuError) as ei:
        _plan(sia, w, monkeypatch, env)
    assert ei.value.code == 1 and needle in ei.value.message
    # left to itself the planner chunks the same grid
    pl = _plan(sia, w, monkeypatch)
    assert pl.kernel == 2 and pl.chunks > 1 and pl.lds_bytes <= 163840


def test_forced_chunk_count_rounds_to_a_realisable_plan(sia, monkeypatch):
    """SDPGPU_WIN_NCH=50 on 125 register blocks: 3 blocks per chunk = 42 chunks (was: no plan, a launch error)."""
    pl = _plan(sia, workloads.target_grid(T=2), monkeypatch, {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "50"})
    assert (pl.chunks, pl.chunk_blocks) == (42, 3)
    pl = _plan(sia, workloads.target_grid(T=2), monkeypatch, {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "9999"})
    assert (pl.chunks, pl.chunk_blocks) == (125, 1)
