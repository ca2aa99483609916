"""What does the K-periods-per-exchange schedule cost a MIDDLE rank in compute?  One GPU plays rank 3 of 8 of the
cfg2 weak-scaling job (8e4 states, a 1e4-state slab) and runs exactly the widened ranges solve_blocked(K) would
give it -- no exchange at all, so the tables hold garbage outside the slab; only the launch durations matter here.
Compared with the plain slab sweep this is the price of the redundant halo states, including what the few extra
tiles do to the number of rounds the launch needs (SDPGPU_DEBUG_PLAN=1 prints the plan of period T)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stochastic_inventory_amd import workloads
from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver

WORLD, RANK = 8, 3
torch.cuda.set_device(0)
w = workloads.cfg2_clsp(S=10000 * WORLD)
desc = w.desc()
desc.rank, desc.world_size = RANK, WORLD
be = GpuSlabBackend(desc, w.pmf)
s = ShardedSolver(be)
assert s.prepare_blocked(8)  # the halo of the widest schedule, before the first run


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def slab_sweep():
    for p in range(w.T, 0, -1):
        be.run_period(p)
    be.finalize()


plain = timed(slab_sweep)
print(f"slab only                       {plain:.3f} ms = {plain / w.T * 1e3:.1f} us/period", flush=True)
for k in (2, 4, 8):
    plan = s.plan_blocks(k)
    if plan is None:
        print(f"K={k}: no bounded footprint"); continue
    blocks, halo = plan
    prog = []
    for t_hi, t_lo, ext in blocks:
        for p in range(t_hi, t_lo - 1, -1):
            _, lo, hi = be.slab(p)
            prog.append((p, max(0, lo - ext[p][0]), min(be.num_states(p), hi + ext[p][1])))

    def widened():
        for p, a, b in prog:
            be.run_period_range(p, a, b)
        be.finalize()

    tk = timed(widened)
    extra = sum(b - a for _, a, b in prog) / sum(be.slab(p)[2] - be.slab(p)[1] for p, _, _ in prog) - 1
    print(f"K={k}: halo {halo:5d}, +{extra * 100:4.1f}% states  {tk:.3f} ms = {tk / w.T * 1e3:.1f} us/period"
          f"  (x{tk / plain:.3f})", flush=True)
be.close()
