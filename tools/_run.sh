SDP_FUZZ_N=1500 timeout -k 10 1000 python -m pytest tests/test_gpu_staff.py -m gpu -x -q -k "random_instances" > gpurun_out/soak_staff.log 2>&1; echo "staff default: $(tail -1 gpurun_out/soak_staff.log)"
python - <<'PY' > gpurun_out/soak_staffwin.log 2>&1
# the window kernel forced on 600 random staff instances
import os, sys
os.environ["SDPGPU_STAFF_PAIR"] = "1"; os.environ["SDPGPU_STAFF_WIN"] = "4"
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import stochastic_inventory_amd as sia
from oracle import staffref
import test_gpu_staff as t
bad = 0
for seed in range(1000, 1600):
    c = t._random_case(seed)
    V, pol, cells = c.oracle_problem(staffref).solve()
    with t._engine(sia, c) as eng:
        eng.solve(sync=True)
        for period in range(1, c.T + 1):
            if not (np.array_equal(eng.values(period), V[period - 1]) and np.array_equal(eng.policy(period), pol[period - 1])):
                bad += 1; print("MISMATCH", seed, period)
print("window kernel forced on 600 random staff instances: mismatches", bad)
PY
tail -2 gpurun_out/soak_staffwin.log
