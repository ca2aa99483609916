"""How much does the per-period exchange cost on the host side?  The GPU box has one GPU, so this runs the
sharded schedule at world_size 1 with the RCCL all-gather FORCED (a 1-rank in-place all-gather moves nothing,
but it pays the whole torch.distributed + RCCL launch path), eagerly and captured in a HIP graph, and
compares with the plain sweep.  What it cannot show is the wire latency of N > 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver
from stochastic_inventory_amd import workloads

w = workloads.cfg2_clsp()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    be = GpuSlabBackend(w.desc(), w.pmf)
    s = ShardedSolver(be)
    s.force_exchange = True
    s.prepare_blocked(8)
    s.force_exchange = False

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    plain = timed(lambda: s.solve())
    print(f"plain sweep (no exchange)            {plain:.3f} ms  = {plain / w.T * 1e3:.1f} us/period", flush=True)
    s.force_exchange = True
    eager_block = timed(lambda: s.solve(overlap=False))
    print(f"forced all-gather, blocking           {eager_block:.3f} ms  = {eager_block / w.T * 1e3:.1f} us/period", flush=True)
    eager = timed(lambda: s.solve(overlap=True))
    print(f"forced all-gather, async + split      {eager:.3f} ms  = {eager / w.T * 1e3:.1f} us/period", flush=True)
    if s.plan_blocks(8) is not None:
        for k in (1, 2, 4, 8):
            tk = timed(lambda: s.solve_blocked(k))
            print(f"forced all-gather, {k} period(s)/wait    {tk:.3f} ms  = {tk / w.T * 1e3:.1f} us/period", flush=True)
    if s.plan_blocks(8) is not None:  # the batched publication path really ran: same tables as a plain sweep?
        import numpy as np
        s.solve_blocked(4)
        torch.cuda.synchronize()
        ref = GpuSlabBackend(w.desc(), w.pmf)
        ref.engine.solve()
        same = all(np.array_equal(be.engine.values(t), ref.engine.values(t)) and
                   np.array_equal(be.engine.policy(t), ref.engine.policy(t)) for t in (1, 2, 3, 17, w.T))
        print("blocked schedule with batched publication == plain sweep:", same, flush=True)
        ref.close()
    if "--graph" in sys.argv:
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            s.solve(overlap=False)
        torch.cuda.synchronize()
        graphed = timed(lambda: g.replay())
        print(f"forced all-gather, one HIP graph      {graphed:.3f} ms  = {graphed / w.T * 1e3:.1f} us/period", flush=True)
        v = be.engine.values(1)
        be2 = GpuSlabBackend(w.desc(), w.pmf)
        be2.engine.solve()
        import numpy as np
        print("graph result == plain result:", bool(np.array_equal(v, be2.engine.values(1))), flush=True)
dist.destroy_process_group()
