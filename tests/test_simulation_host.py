"""Host-side pieces of the simulation mirror that need no GPU: the reproducible latin-hypercube sampler
(Sampling.java:86-103) and the Math.round of the samples."""
import numpy as np

from stochastic_inventory_amd import pmf as PM
from stochastic_inventory_amd.simulation import Sampling, round_demands


def test_lhs_has_one_sample_per_stratum_and_is_reproducible():
    dists = [PM.PoissonDist(10.0), PM.NormalDist(20.0, 5.0)]
    n = 400
    a = Sampling(seed=5).generateLHSamples(dists, n)
    b = Sampling(seed=5).generateLHSamples(dists, n)
    assert np.array_equal(a, b) and a.shape == (n, 2)
    # stratification: the normal column, mapped back through the cdf, hits every stratum [j/n, (j+1)/n) once
    u = np.sort([dists[1].cdf(v) for v in a[:, 1]])
    assert np.all(np.floor(u * n + 1e-9).astype(int) == np.arange(n))
    # the columns are shuffled independently (the reference shuffles column by column)
    assert abs(np.corrcoef(a[:, 0], a[:, 1])[0, 1]) < 0.2


def test_round_demands_is_java_round():
    x = np.array([[0.5, 1.5, 2.4999, -0.5]])
    assert round_demands(x).tolist() == [[1.0, 2.0, 2.0, 0.0]]
