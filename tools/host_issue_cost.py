import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
w = workloads.cfg2_clsp()
d = w.desc(); d.rank, d.world_size = 1, 4   # a middle rank of 4: real interior/boundary split
eng = sia.SdpEngine(d, w.pmf)
for mode in ("whole", "split"):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for period in range(w.T, 0, -1):
            if mode == "whole" or period == w.T:
                eng.run_period(period)
            else:
                eng.run_period_part(period, 1); eng.run_period_part(period, 2)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
    print(f"{mode}: host issue {(t1-t0)/w.T*1e6:.1f} us/period, total {(t2-t0)/w.T*1e6:.1f} us/period")
