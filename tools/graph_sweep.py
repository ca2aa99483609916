"""Does capturing the 52 period launches of a configs[1] sweep in one HIP graph shorten it?  (The launches are
already issued back to back on one stream; what a graph could remove is per-launch dispatch overhead.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stochastic_inventory_amd import workloads
from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver

torch.cuda.set_device(0)
w = workloads.cfg2_clsp()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    be = GpuSlabBackend(w.desc(), w.pmf)
    s = ShardedSolver(be)

    def timed(fn, reps=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    plain = timed(lambda: s.solve())
    print(f"eager sweep      {plain:.4f} ms = {plain / w.T * 1e3:.2f} us/period", flush=True)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=stream):
        s.solve()
    torch.cuda.synchronize()
    graphed = timed(lambda: g.replay())
    print(f"one HIP graph    {graphed:.4f} ms = {graphed / w.T * 1e3:.2f} us/period", flush=True)
    import numpy as np
    v = be.engine.values(1).copy()
    s.solve(); torch.cuda.synchronize()
    print("graph result == eager result:", bool(np.array_equal(v, be.engine.values(1))), flush=True)
