"""N > 1 host logic on CPU: world_size-2 (and 3) `gloo` runs of the very ShardedSolver class the GPU
bench uses, with a test double in place of the HIP engine.  The double computes its slab with the
oracle (allowed here: tests/), so what is under test is the sharding itself -- slab bounds, padding,
the in-place all-gather offsets, policy slabs -- not the arithmetic."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, case_name, out_dir, blocked_k=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd.sharded import ShardedSolver, SlabBackend

    w = getattr(cases, case_name)()
    desc = w.desc()
    desc.rank, desc.world_size = rank, world
    geom = sia.SdpEngine(desc, w.pmf, w.overhead())  # layout/slab queries only: no GPU work on CPU
    P = sdpref.Problem(w.desc(), w.pmf, w.overhead())

    class OracleSlab(SlabBackend):
        T = w.T

        def __init__(self):
            pads = [geom.slab(p)[0] for p in range(1, w.T + 1)]
            if blocked_k and len(set(pads)) == 1:  # one arena [period][padded row], like the GPU backend
                self.arena = torch.full((w.T, pads[0]), float("nan"), dtype=torch.float64)
                self.tables = {p: self.arena[p - 1] for p in range(1, w.T + 1)}
            else:
                self.arena = None
                self.tables = {p: torch.full((pads[p - 1],), float("nan"), dtype=torch.float64) for p in range(1, w.T + 1)}
            self.policy = {}

        def table_block(self, p_lo, p_hi):
            return None if self.arena is None else self.arena[p_lo - 1: p_hi]

        def slab(self, period):
            return geom.slab(period)

        def table(self, period):
            return self.tables[period]

        def run_period(self, period):
            pad, lo, hi = geom.slab(period)
            S = geom.num_states(period)
            v_next = self.tables[period + 1][:geom.num_states(period + 1)].numpy() if period < w.T else None
            if v_next is not None:
                assert not np.isnan(v_next).any(), "a slab of V_{t+1} was not gathered"
            v = np.full(S, np.nan)
            pol = np.zeros(S, dtype=np.int32)
            P.period(period, v_next, lo, hi, 1, v, pol)
            self.tables[period][lo:hi] = torch.from_numpy(v[lo:hi])
            self.policy[period] = pol[lo:hi].copy()

        # -- the widened-slab interface of solve_blocked --
        def footprint(self, period):
            return geom.footprint(period)

        def num_states(self, period):
            return geom.num_states(period)

        def set_halo(self, halo):
            self.halo = halo

        def run_period_range(self, period, a, b):
            pad, lo, hi = geom.slab(period)
            assert lo - self.halo <= a <= b <= hi + self.halo or a == b
            S = geom.num_states(period)
            v_next = self.tables[period + 1][:geom.num_states(period + 1)].numpy() if period < w.T else None
            v = np.full(S, np.nan)
            pol = np.zeros(S, dtype=np.int32)
            if b > a:
                P.period(period, v_next, a, b, 1, v, pol)  # a NaN it reads (an un-gathered slab) poisons the result
                assert not np.isnan(v[a:b]).any(), "the widened slab read a part of V_{t+1} that was never valid here"
                self.tables[period][a:b] = torch.from_numpy(v[a:b])
            self.policy[period] = pol[lo:hi].copy()

    class LateWork:
        """An exchange that happens as late as the schedule allows: at wait()."""

        def __init__(self, fn):
            self.fn, self.done = fn, False

        def wait(self):
            if not self.done:
                self.fn()
                self.done = True

    class LateSolver(ShardedSolver):
        """Worst case for solve_blocked: every all-gather completes only when somebody waits for it, so a period
        that read a row before its block boundary was waited for would read NaNs."""

        def exchange(self, period, async_op=False):
            if not async_op:
                return super().exchange(period)
            return LateWork(lambda: ShardedSolver.exchange(self, period))

        def exchange_many(self, p_lo, p_hi):  # the batched publication, also as late as possible
            if p_hi < p_lo:
                return None

            def late():
                done = ShardedSolver.exchange_many(self, p_lo, p_hi)
                if done is not None:
                    done()
            return late

    be = OracleSlab()
    solver = LateSolver(be) if blocked_k else ShardedSolver(be)
    if blocked_k:
        assert solver.prepare_blocked(blocked_k)
        solver.solve_blocked(blocked_k)
    else:
        solver.solve()
    # V_1 is not exchanged by solve(); gather it here only to compare the whole table
    solver.exchange(1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             **{f"v{p}": be.tables[p].numpy() for p in range(1, w.T + 1)},
             **{f"p{p}": be.policy[p] for p in range(1, w.T + 1)})
    geom.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case_name", [(2, "f1_small"), (2, "f3_tenths"), (3, "f2_unclamped"), (2, "f5_cash_leadtime")])
def test_sharded_solver_gloo(tmp_path, oracle, world, case_name):
    import cases
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, case_name, str(tmp_path)), nprocs=world, join=True)
    w = getattr(cases, case_name)()
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for p in range(1, w.T + 1):
        S = len(V[p - 1])
        for r in range(world):
            assert np.array_equal(ranks[r][f"v{p}"][:S], V[p - 1]), (p, r)  # every rank holds the full table
        assert np.array_equal(np.concatenate([ranks[r][f"p{p}"] for r in range(world)]), pol[p - 1])


@pytest.mark.parametrize("world,case_name,k", [(2, "f1_small", 2), (3, "f1_small", 3), (3, "f1_clsp_main", 2),
                                               (4, "f1_clsp_main", 4), (2, "f1_gapped", 3), (8, "f1_clsp_main", 3)])
def test_blocked_schedule_gloo(tmp_path, oracle, world, case_name, k):
    """K periods per exchange (ShardedSolver.solve_blocked): every period of a block runs on a widened slab, the
    all-gathers only publish rows.  Same tables as the oracle's single sweep on every rank."""
    import cases
    port = 29500 + (os.getpid() % 2000) + 10 * world + k
    mp.spawn(_worker, args=(world, port, case_name, str(tmp_path), k), nprocs=world, join=True)
    w = getattr(cases, case_name)()
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for p in range(1, w.T + 1):
        S = len(V[p - 1])
        for r in range(world):
            assert np.array_equal(ranks[r][f"v{p}"][:S], V[p - 1]), (p, r)
        assert np.array_equal(np.concatenate([ranks[r][f"p{p}"] for r in range(world)]), pol[p - 1])


def _native_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    import stochastic_inventory_amd as sia
    from stochastic_inventory_amd.sharded import init_native_comm
    w = cases.f1_small()
    desc = w.desc()
    desc.rank, desc.world_size = rank, world
    with sia.SdpEngine(desc, w.pmf) as eng:
        got = init_native_comm(eng)  # no GPU here: sdpgpu_comm_init fails on every rank -- and every rank must say so
        with open(os.path.join(out_dir, f"native{rank}.txt"), "w") as f:
            f.write(f"{got}|{init_native_comm.last_error}")
    dist.barrier()
    dist.destroy_process_group()


def test_native_comm_setup_agrees_across_ranks_without_a_gpu(tmp_path):
    """sharded.init_native_comm on CPU, world 3: librccl loads and rank 0's unique id reaches every rank (phase 1), the
    communicator itself needs a device (phase 2 fails: the library has no CPU path), and the ranks AGREE on the failure
    instead of leaving each other inside a collective -- which is what lets bench.py fall back together."""
    world = 3
    port = 29500 + (os.getpid() % 2000) + 77
    mp.spawn(_native_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got, err = open(os.path.join(str(tmp_path), f"native{r}.txt")).read().split("|", 1)
        assert got == "False"
        assert "HIP" in err or "device" in err.lower()
