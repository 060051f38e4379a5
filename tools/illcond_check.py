import sys, numpy as np
sys.path.insert(0, '/root/repo')
from ml_amd import _lib
from oracle import oracle_ctypes as orc
rng = np.random.default_rng(5)
d, K, n = 8, 4, 4000
for sigma in (1.0, 1e-2, 1e-4, 1e-6):
    means = 10.0 * rng.standard_normal((K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + sigma * rng.standard_normal((n, d)))
    mu0 = means + 0.1 * sigma * rng.standard_normal((K, d))
    S0 = np.stack([np.eye(d) * sigma ** 2] * K)
    pi0 = np.full(K, 1.0 / K)
    ctx = _lib.Context(); dt = _lib.Data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(pi0, mu0, S0)
    em = orc.EM(K); em.set_parameters(mu0, S0, pi0); em.expectation_step(X); em.maximisation_step(X)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    print(f"sigma={sigma:g}: LL rel {abs(ll-em.log_likelihood)/abs(em.log_likelihood):.2e}  mu rel {rel(mu1, em.means):.2e}  cov rel {rel(S1, em.covariances):.2e}")
    dt.close(); ctx.close()
