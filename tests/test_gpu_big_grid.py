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


def _every_state_of_period_1_and_bands_of_period_2(eng, P, threads):
    """Period 1 (the one with a future term), EVERY state, fed the GPU's own V_2; period 2 (the last: no gather) on three
    contiguous bands -- both ends of the grid and its middle, a twentieth each.  Returns the cells the oracle evaluated.
    (The oracle's share of this file is what the GPU test run's wall-clock is made of: 2e9 cells/s on 16 host threads.)"""
    import numpy as np
    v2, p2 = eng.values(2), eng.policy(2)
    S = len(v2)
    cells = 0
    for lo, hi in ((0, S // 20), (S // 2, S // 2 + S // 20), (S - S // 20, S)):
        ov, oa, c = P.period(2, None, lo, hi, nthreads=threads)
        assert np.array_equal(v2[lo:hi], ov[lo:hi]) and np.array_equal(p2[lo:hi], oa[lo:hi]), ("period 2", lo, hi)
        cells += c
    ov1, oa1, c1 = P.period(1, v2, nthreads=threads)
    assert np.array_equal(eng.values(1), ov1) and np.array_equal(eng.policy(1), oa1), "period 1"
    return cells + c1


def test_target_grid_full_tables():
    """The north-star target grid (1e6 states x 500 actions x 200 demands), the bench headline's workload: EVERY state of
    the period with a future term (1e6 states, 1e11 cells), fed the GPU's own V_2, and three bands of the last period,
    against the oracle -- values and policy indices bit-identical; the clamp edges of the grid are part of the tables."""
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
        assert _every_state_of_period_1_and_bands_of_period_2(eng, P, threads) == 10 ** 11 + 3 * 5 * 10 ** 9


def test_cfg3_full_tables():
    """configs[2] at its full cash width, action and demand counts (5000 cash points, <= 300 actions, 150 demands) on 100 of its
    200 inventory rows: EVERY state of period 1 -- the diagonal kernel (cash_diag_kernel), fed the GPU's own V_2 -- and three
    bands of period 2 (the uniform-shift kernel) against the oracle: values and policy indices bit-identical, 2.4e10 cells.
    (The full 200 rows are covered by sampled states, test_full_size_config_sampled_parity, and by bench.py's parity gate.)"""
    import numpy as np
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd import workloads
    w = workloads.cfg3_cash(T=2, NX=100)
    threads = min(os.cpu_count() or 1, 16)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        eng.solve()
        P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
        assert _every_state_of_period_1_and_bands_of_period_2(eng, P, threads) > 2e10


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


def test_cfg3t_full_tables():
    """configs[2]'s family at the cash axis and quantum of the reference's own CashConstraint.main (20001 cash points in tenths,
    101 actions, 25 demands) on 251 of its 501 inventory rows: EVERY state of period 1, fed the GPU's own V_2, and three bands
    of period 2 against the oracle -- the cash row pair kernel (two points per lane, two tiles per wave, uniform-key trips,
    XCD cash bands) as bench.py's cfg3t entry runs it; 1.4e10 cells."""
    import numpy as np
    import stochastic_inventory_amd as sia
    from oracle import sdpref
    from stochastic_inventory_amd import workloads
    w = workloads.cfg3_tenths(T=2, NX=251)
    threads = min(os.cpu_count() or 1, 16)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        eng.solve()
        P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
        assert _every_state_of_period_1_and_bands_of_period_2(eng, P, threads) > 1.2e10 and eng.stats().kernel_used == 2
