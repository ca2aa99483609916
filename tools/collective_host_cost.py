import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver
from stochastic_inventory_amd import workloads
w = workloads.cfg2_clsp()
be = GpuSlabBackend(w.desc(), w.pmf)
s = ShardedSolver(be)
s.force_exchange = True
s.prepare_blocked(8)
for name, fn in (("blocking", lambda: s.solve(overlap=False)), ("blocked8", lambda: s.solve_blocked(8)), ("blocked2", lambda: s.solve_blocked(2))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host issue {(t1-t0)/w.T*1e6:.1f} us/period, total {(t2-t0)/w.T*1e6:.1f} us/period")
# cost of one async all_gather call + wait on the host
full = be.table(2); pad = full.numel(); 
t0 = time.perf_counter()
ws = [dist.all_gather_into_tensor(full, full[0:pad], async_op=True) for _ in range(200)]
t1 = time.perf_counter()
for x in ws: x.wait()
t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"async all_gather: issue {(t1-t0)/200*1e6:.1f} us, wait {(t2-t1)/200*1e6:.1f} us, drain {(t3-t2)*1e6:.0f} us total")
dist.destroy_process_group()
