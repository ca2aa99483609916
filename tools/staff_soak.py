"""Soak of the staff kernels' forms (GPU box): seeded random instances (tests/test_gpu_staff.py: _random_case) under the launcher's
own choice and with each form forced, every table against oracle/staffref.c.   python tools/staff_soak.py [first] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stochastic_inventory_amd as sia
from oracle import staffref
import test_gpu_staff as tg

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
forms = [("auto", {}), ("window S=4", {"SDPGPU_STAFF_PAIR": "1", "SDPGPU_STAFF_WIN": "4"}),
         ("window S=2", {"SDPGPU_STAFF_PAIR": "1", "SDPGPU_STAFF_WIN": "2"}), ("lanes = actions", {"SDPGPU_STAFF_LANES": "1"}),
         ("window S=4, staged only", {"SDPGPU_STAFF_PAIR": "1", "SDPGPU_STAFF_WIN": "4", "SDPGPU_STAFF_UNI": "0"})]
bad = 0
for seed in range(first, first + count):
    c = tg._random_case(seed)
    V, pol, cells = c.oracle_problem(staffref).solve()
    for label, env in forms:
        for k in ("SDPGPU_STAFF_PAIR", "SDPGPU_STAFF_WIN", "SDPGPU_STAFF_LANES", "SDPGPU_STAFF_UNI"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with tg._engine(sia, c) as eng:
            eng.solve(sync=True)
            ok = eng.stats().cells_evaluated == cells and all(
                np.array_equal(eng.values(t), V[t - 1]) and np.array_equal(eng.policy(t), pol[t - 1]) for t in range(1, c.T + 1))
        if not ok:
            bad += 1
            print("MISMATCH", seed, label, flush=True)
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} instances x {len(forms)} forms, mismatches so far: {bad}", flush=True)
print("staff soak:", count, "instances,", len(forms), "forms each, mismatches:", bad)
sys.exit(1 if bad else 0)
