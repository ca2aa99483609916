"""RCCL sanity at world_size 1 (the GPU box has one GPU): init, in-place all_gather_into_tensor on a
slice of its own output -- the exact call ShardedSolver.exchange makes -- and the bench path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
full = torch.arange(1024, dtype=torch.float64, device="cuda")
shard = full[0:1024]
dist.all_gather_into_tensor(full, shard)
torch.cuda.synchronize()
assert torch.equal(full.cpu(), torch.arange(1024, dtype=torch.float64))
from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver
from stochastic_inventory_amd import workloads
w = workloads.cfg2_clsp(T=4)
be = GpuSlabBackend(w.desc(), w.pmf)
s = ShardedSolver(be); s.world = 1
s.solve(); torch.cuda.synchronize()
# force the exchange code path at world 1
s.world = 1
pad, lo, hi = be.slab(2)
dist.all_gather_into_tensor(be.table(2), be.table(2)[0:pad])
torch.cuda.synchronize()
print("nccl world-1 ok", float(be.table(1)[5000].item()))
dist.destroy_process_group()
