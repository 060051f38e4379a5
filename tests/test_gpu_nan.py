"""A component with NaN parameters must poison the log-likelihood, as in the reference (ML/EM.cpp:205-218: exp(NaN) = NaN
enters the row sum, log(NaN) the mean) and in the oracle -- not vanish as a zero responsibility (exp_nonpos once clamped a NaN
to -800). A poisoned fit never reports convergence (ML/EM.cpp:161-168: `|dLL| < tol` is false for a NaN)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


def _problem(d, K, n, seed):
    rng = np.random.default_rng(seed)
    means = 3.0 * rng.standard_normal((K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)))
    return X, means + 0.1 * rng.standard_normal((K, d))


@pytest.mark.parametrize("d,K,n,diagonal", [
    (32, 64, 6000, False),    # matrix-core E-step, self-normalising statistics kernel
    (16, 40, 6000, False),    # matrix-core E-step with its own log-sum-exp
    (8, 5, 5000, False),      # scalar-fed E-step
    (4, 3, 5000, False),      # fused small-shape kernel
    (48, 4, 3000, False),
    (16, 16, 6000, True),     # diagonal kernel
])
@pytest.mark.parametrize("what", ["mean", "covariance"])
def test_nan_parameters_poison_the_log_likelihood(ctx, oracle, d, K, n, diagonal, what):
    from ml_amd import _lib
    X, mu0 = _problem(d, K, n, 7 * d + K)
    dt = _lib.Data(ctx, X)
    var = np.var(X, axis=0)
    S0 = np.tile(var, (K, 1)) if diagonal else np.stack([np.diag(var)] * K)
    pi0 = np.full(K, 1.0 / K)
    if what == "mean":
        mu0[K - 1, 0] = np.nan
    elif diagonal:
        S0[K - 1, 0] = np.nan
    else:
        S0[K - 1, 0, 0] = np.nan
    step = dt.em_step_diag if diagonal else dt.em_step
    ll, pi1, mu1, S1 = step(pi0, mu0, S0)
    assert np.isnan(ll)
    assert np.isnan(mu1).any()
    # the oracle agrees
    em = oracle.EM(K)
    if diagonal:
        em.set_covariance_type("diag")
    em.set_parameters(mu0, np.stack([np.diag(v) for v in S0]) if diagonal else S0, pi0)
    em.expectation_step(X)
    assert np.isnan(em.log_likelihood)
    # the loop in one call: never converges, runs its max_steps
    steps, conv, ll_it, *_ = dt.em_iterate(pi0, mu0, S0, 4, 1e-6, 1e-6, diagonal)
    assert not conv and steps == 4 and np.isnan(ll_it)
    dt.close()


@pytest.mark.parametrize("d,K,n", [
    (16, 40, 6000),      # matrix-core E-step with its own log-sum-exp
    (32, 64, 6000),      # ... without (the statistics kernel normalises)
    (8, 5, 5000),        # scalar-fed E-step
    (4, 3, 5000),        # fused small-shape kernel
    (150, 3, 1500),      # plain tier (d > 128)
])
@pytest.mark.parametrize("zeros", [(0,), (0, 1), (2,)])
def test_zero_mixing_weight_stays_finite_like_the_reference(ctx, oracle, d, K, n, zeros):
    """pi_k = 0 multiplies column k by 0 in the reference (ML/EM.cpp:209): the log-likelihood and the other components'
    responsibilities stay finite. In the log domain that component is lw = -inf, which must not turn the online log-sum-exp into
    -inf - -inf = NaN when it comes FIRST."""
    from ml_amd import _lib
    X, mu0 = _problem(d, K, n, 3 * d + K)
    dt = _lib.Data(ctx, X)
    S0 = np.stack([np.diag(np.var(X, axis=0))] * K)
    pi0 = np.full(K, 1.0)
    pi0[list(zeros)] = 0.0
    pi0 /= pi0.sum()
    ll = dt.em_expectation(pi0, mu0, S0)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert np.isfinite(ll) and abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    R = dt.em_responsibilities(K)
    assert np.all(R[:, list(zeros)] == 0.0)
    assert np.max(np.abs(R - em.responsibilities)) < 1e-12
    em.calculate_labels()
    assert np.array_equal(dt.em_labels(K), em.labels)
    ll_step = dt.em_step(pi0, mu0, S0)[0]                     # the route mlhip_em_iterate takes (fused / self-normalising kernels)
    assert abs(ll_step - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    dt.close()
