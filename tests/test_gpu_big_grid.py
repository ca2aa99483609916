"""configs[4] at full width on one GPU (1e8 states x 500 actions x 200 demands, 2 periods = 2e13 cells):
the oracle cannot sweep that, so 3000 sampled states per period are compared with the oracle fed the GPU's
own V_{t+1} -- bit for bit -- plus the clamp edges of the grid."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_1e8_state_grid_sampled_parity():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_grid_check.py"), "100000000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("bit-identical") == 2


def test_1e7_state_pipeline_grid_sampled_parity():
    """configs[3] at full size: (x, q1, q2) = 250 x 200 x 200 states x 200 actions x 100 demands, 3 periods."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pipeline_grid_check.py"), "3"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("bit-identical") == 3


@pytest.mark.parametrize("name,periods", [("cfg3", 4), ("cfg4", 4)])
def test_full_size_config_sampled_parity(name, periods):
    """configs[2] (1e6 cash states x <=300 actions x 150 demands, 6 periods) and configs[3] in the reference's
    (period, x, preQ) shape (50 periods x 1000 x 200) at full size: sampled states of four periods against the
    oracle fed the GPU's own V_{t+1}."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sampled_grid_check.py"), name, "1500"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("bit-identical") == periods


def test_target_grid_full_tables_two_periods():
    """The north-star target grid (1e6 states x 500 actions x 200 demands), the bench headline's workload: EVERY state
    of two periods (2e6 state-periods, 2e11 cells) against the oracle -- the period with a future term fed the GPU's own
    V_2 -- values and policy indices bit-identical; the clamp edges of the grid are part of the table."""
    import numpy as np
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd import workloads
    w = workloads.target_grid(T=2)
    threads = min(os.cpu_count() or 1, 16)
    with sia.SdpEngine(w.desc(), w.pmf) as eng:
        eng.solve()
        assert eng.stats().cells_evaluated == 2 * 10 ** 11
        P = sdpref.Problem(w.desc(), w.pmf)
        v2, p2 = eng.values(2), eng.policy(2)
        ov2, oa2, c2 = P.period(2, None, nthreads=threads)
        assert np.array_equal(v2, ov2) and np.array_equal(p2, oa2) and c2 == 10 ** 11
        ov1, oa1, _ = P.period(1, v2, nthreads=threads)
        assert np.array_equal(eng.values(1), ov1) and np.array_equal(eng.policy(1), oa1)


def test_cfg3_full_tables_two_periods():
    """configs[2] at its full size (200 x 5000 states, <= 300 actions, 150 demands): EVERY state of two periods -- period 2
    on the uniform-shift kernel, period 1 on the diagonal kernel (cash_diag_kernel), fed the GPU's own V_2 -- against the
    oracle: values and policy indices bit-identical, 8.7e10 cells."""
    import numpy as np
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd import workloads
    w = workloads.cfg3_cash(T=2)
    threads = min(os.cpu_count() or 1, 16)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        eng.solve()
        P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
        v2, p2 = eng.values(2), eng.policy(2)
        ov2, oa2, c2 = P.period(2, None, nthreads=threads)
        assert np.array_equal(v2, ov2) and np.array_equal(p2, oa2)
        ov1, oa1, c1 = P.period(1, v2, nthreads=threads)
        assert np.array_equal(eng.values(1), ov1) and np.array_equal(eng.policy(1), oa1)
        assert eng.stats().cells_evaluated == c1 + c2


def test_cfg4_full_tables_three_periods():
    """configs[3] in the reference's (period, x, preQ) shape at its full width (1000 x 200 states, 200 actions, 100 demands),
    three periods: EVERY state of every period against the oracle (1.2e10 cells) -- the row-window kernel with four states
    per lane, wave-private row staging and 16-byte LDS reads, as bench.py's cfg4 entry runs it."""
    import numpy as np
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd import workloads
    w = workloads.cfg4_leadtime(T=3)
    threads = min(os.cpu_count() or 1, 16)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        eng.solve()
        V, pol, cells = sdpref.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=threads)
        assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2
        for period in range(1, 4):
            assert np.array_equal(eng.values(period), V[period - 1]) and np.array_equal(eng.policy(period), pol[period - 1]), period


def test_cfg3t_full_tables_two_periods():
    """configs[2]'s family at the size and cash quantum of the reference's own CashConstraint.main (501 x 20001 states, cash
    in tenths, 101 actions, 25 demands): EVERY state of two periods against the oracle -- the cash row pair kernel (two points
    per lane, two tiles per wave, uniform-key trips, XCD cash bands) as bench.py's cfg3t entry runs it; 5e10 cells."""
    import numpy as np
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd import workloads
    w = workloads.cfg3_tenths(T=2)
    threads = min(os.cpu_count() or 1, 16)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        eng.solve()
        P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
        v2, p2 = eng.values(2), eng.policy(2)
        ov2, oa2, c2 = P.period(2, None, nthreads=threads)
        assert np.array_equal(v2, ov2) and np.array_equal(p2, oa2)
        ov1, oa1, c1 = P.period(1, v2, nthreads=threads)
        assert np.array_equal(eng.values(1), ov1) and np.array_equal(eng.policy(1), oa1)
        assert eng.stats().cells_evaluated == c1 + c2 and eng.stats().kernel_used == 2
