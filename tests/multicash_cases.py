"""Instances of the two-product cash-constrained family (CashRecursionMulti over MultiItemCash's lambdas)."""
import numpy as np
from scipy import stats


def normal_joint_pmf(means, sigmas, q=0.999):
    """GetPmfMulti.getPmf for two NormalDists (GetPmfMulti.java:46-69): integer supports between the (int)-truncated
    quantiles, cell probabilities from the cdf, divided by (2q - 1)^2.  scipy stands in for SSJ (parity unpinned at
    the pmf boundary; the recursion takes the list as input)."""
    lb = [int(stats.norm.ppf(1 - q, m, s)) for m, s in zip(means, sigmas)]
    ub = [int(stats.norm.ppf(q, m, s)) for m, s in zip(means, sigmas)]
    rows = []
    psum = (2 * q - 1) * (2 * q - 1)
    for i in range(ub[0] - lb[0] + 1):
        for j in range(ub[1] - lb[1] + 1):
            d1, d2 = lb[0] + i, lb[1] + j
            p = ((stats.norm.cdf(d1 + 0.5, means[0], sigmas[0]) - stats.norm.cdf(d1 - 0.5, means[0], sigmas[0])) *
                 (stats.norm.cdf(d2 + 0.5, means[1], sigmas[1]) - stats.norm.cdf(d2 - 0.5, means[1], sigmas[1])) / psum)
            rows.append([float(d1), float(d2), float(p)])
    return np.array(rows)


def main_instance():
    """MultiItemCash.main as it stands (MultiItemCash.java:28-57): price {4, 50}, variCost {2, 4}, iniCash 100,
    demand means {5, 6} for both products with coefficient of variation 0.25, T = 2, Qbound 100."""
    demand, coe = [[5, 6], [5, 6]], [0.25, 0.25]
    pmf = [normal_joint_pmf([demand[0][t], demand[1][t]], [coe[0] * demand[0][t], coe[1] * demand[1][t]]) for t in range(2)]
    return dict(T=2, q_bound=100, price=[4, 50], vari_cost=[2, 4], sal_price=[1, 1], ini_cash=100, ini_i1=0, ini_i2=0,
                min_inventory=0, max_inventory=200, min_cash=0, max_cash=10000, discount=1, pmf=pmf)


def random_instance(seed):
    rng = np.random.default_rng(seed)
    T = int(rng.integers(1, 4))
    q_bound = int(rng.integers(2, 7 if T == 3 else 12))
    pmf = []
    for _ in range(T):
        n1, n2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        v1 = np.sort(rng.choice(np.arange(0, 9), size=n1, replace=False))
        v2 = np.sort(rng.choice(np.arange(0, 9), size=n2, replace=False))
        p = rng.random((n1, n2)) + 0.05
        p /= p.sum()
        pmf.append(np.array([[float(a), float(b), float(p[i, j])] for i, a in enumerate(v1) for j, b in enumerate(v2)]))
    return dict(T=T, q_bound=q_bound, price=[float(rng.integers(2, 12)), float(rng.integers(2, 30)) / 2],
                vari_cost=[float(rng.integers(1, 6)) / 2, float(rng.integers(1, 9)) / 2],
                sal_price=[float(rng.integers(0, 3)) / 2, float(rng.integers(0, 3)) / 4],
                ini_cash=float(rng.integers(0, 40)), ini_i1=float(rng.integers(0, 4)), ini_i2=float(rng.integers(0, 4)),
                min_inventory=0, max_inventory=float(rng.integers(3, 12)), min_cash=0, max_cash=float(rng.integers(30, 200)),
                discount=float(rng.choice([1.0, 0.95])), pmf=pmf)


def poisson_joint_pmf(means, q=0.99):
    """GetPmfMulti.getPmf for two PoissonDists (GetPmfMulti.java:107-138)."""
    lb = [int(stats.poisson.ppf(1 - q, m)) for m in means]
    ub = [int(stats.poisson.ppf(q, m)) for m in means]
    psum = (2 * q - 1) * (2 * q - 1)
    return np.array([[float(lb[0] + i), float(lb[1] + j),
                      float(stats.poisson.pmf(lb[0] + i, means[0]) * stats.poisson.pmf(lb[1] + j, means[1]) / psum)]
                     for i in range(ub[0] - lb[0] + 1) for j in range(ub[1] - lb[1] + 1)])


def xr_main_instance(q_bound=50, q=0.99):
    """MultiItemCashXR.main as it stands (MultiItemCashXR.java:41-78): price {5, 10}, variCost {1, 2}, salvage half the
    cost, Poisson demands with means {20, 10}, T = 2, Qbound 50, truncation quantile 0.99, iniCash 0, no deposit."""
    pmf = [poisson_joint_pmf([20, 10], q) for _ in range(2)]
    return dict(T=2, q_bound=q_bound, price=[5, 10], vari_cost=[1, 2], sal_price=[0.5, 1.0], ini_cash=0, ini_i1=0,
                ini_i2=0, min_inventory=0, max_inventory=200, min_cash=0, max_cash=10000, discount=1, pmf=pmf)


def xr_random_instance(seed):
    kw = random_instance(seed)
    rng = np.random.default_rng(1000 + seed)
    # R = cash + variCost . x of the period-1 state
    kw["ini_cash"] = kw["ini_cash"] + kw["vari_cost"][0] * kw["ini_i1"] + kw["vari_cost"][1] * kw["ini_i2"]
    if kw["T"] == 3:
        kw["q_bound"] = min(kw["q_bound"], 4)
    return float(rng.choice([0.0, 0.05])), kw
