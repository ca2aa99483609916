"""Frozen tables (tests/golden/make_golden.py): the oracle must keep reproducing them (CPU) and the
HIP path must match them bit for bit (GPU)."""
import os

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(name):
    return np.load(os.path.join(HERE, "golden", f"tables_{name}.npz"))


@pytest.mark.parametrize("make", cases.ALL, ids=lambda f: f.__name__)
def test_oracle_reproduces_golden(oracle, make):
    w = make()
    g = _golden(w.name)
    V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    assert cells == int(g["cells"])
    for t in range(w.T):
        assert np.array_equal(V[t], g[f"v{t + 1}"])
        assert np.array_equal(pol[t], g[f"p{t + 1}"].astype(np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("make", cases.ALL, ids=lambda f: f.__name__)
def test_gpu_matches_golden(sia, make):
    w = make()
    g = _golden(w.name)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        eng.solve()
        for t in range(w.T):
            assert np.array_equal(eng.values(t + 1), g[f"v{t + 1}"])
            assert np.array_equal(eng.policy(t + 1), g[f"p{t + 1}"].astype(np.int32))
