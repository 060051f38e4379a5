"""Closing arithmetic of the M-step on the device for 64 < d <= 1024 (em_close_big.hip: panelled Cholesky factorization and inverse in
global memory) -- EM::maximisation_step's tail and EM::process_covariances (ML/EM.cpp:242, 250-257, 274-287) -- against the HOST
closing (MLHIP_DEVICE_CLOSE=0: host/em_math.cpp, the arithmetic the parity tests pin to the oracle). Every entry is formed by the same
operations in the same order on both sides, so mixing weights, means and covariances of ONE iteration are the same bits, and the
second iteration's log-likelihood -- which sees the factor, its inverse and sum log L_jj through the E-step's records -- agrees to the
ulp of log()."""
import os

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
    means = 2.0 * rng.standard_normal((K, d))
    A = rng.standard_normal((K, d, d)) / np.sqrt(d)                      # correlated components: full factors, not near-diagonal ones
    comp = rng.integers(0, K, n)
    Z = rng.standard_normal((n, d))
    X = means[comp] + Z + 0.5 * np.einsum("nij,nj->ni", A[comp], Z)
    mu0 = means + 0.1 * rng.standard_normal((K, d))
    return np.ascontiguousarray(X), mu0


def _iterate(dt, pi0, mu0, S0, steps, device_close, device_records=None):
    """em_iterate with the closing arithmetic on the device or on the host; `device_records`: where the records of the FIRST E-step
    are factored (default: where the closing runs)."""
    keep = {k: os.environ.get(k) for k in ("MLHIP_DEVICE_CLOSE", "MLHIP_DEVICE_RECORDS")}
    os.environ["MLHIP_DEVICE_CLOSE"] = "1" if device_close else "0"
    if device_records is not None:
        os.environ["MLHIP_DEVICE_RECORDS"] = "1" if device_records else "0"
    try:
        return dt.em_iterate(pi0, mu0, S0, steps, 0.0, 0.0, False)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("d,K,n", [
    (65, 3, 3000),        # first dimension of the tier: three panels, the last one a single column
    (72, 5, 4000),
    (96, 2, 3000),        # whole panels only
    (100, 3, 4000),
    (128, 4, 5000),       # last dimension of the matrix-core E-step's 4x4-block records
    (129, 2, 3000),       # first dimension of the packed-triangle records (big_dim.hip)
    (200, 3, 3000),
    (256, 2, 3000),
    (300, 1, 2500),
    (512, 2, 4000),
])
def test_device_closing_equals_host_closing(ctx, d, K, n):
    from ml_amd import _lib
    X, mu0 = _problem(d, K, n, 7 * d + K)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    S0, pi0 = np.stack([cov] * K), np.full(K, 1.0 / K)
    # one iteration from the SAME first records (the host's): the new parameters involve no library function -- the same bits
    _, _, ll_h, pi_h, mu_h, S_h, _ = _iterate(dt, pi0, mu0, S0, 1, False, device_records=False)
    _, _, ll_d, pi_d, mu_d, S_d, _ = _iterate(dt, pi0, mu0, S0, 1, True, device_records=False)
    assert ll_d == ll_h
    assert np.array_equal(pi_d, pi_h) and np.array_equal(mu_d, mu_h) and np.array_equal(S_d, S_h)
    # the first records factored on the device (launch_em_records_big) against the host builders: the first log-likelihood sees W and
    # sum log L_jj of the GIVEN covariances
    _, _, ll_r, *_ = _iterate(dt, pi0, mu0, S0, 1, True, device_records=True)
    assert abs(ll_r - ll_h) <= 1e-14 * abs(ll_h)
    # three iterations: the records built on the device (W = L^-1, sum log L_jj, the mean) drive the second and third E-step
    _, _, _, pi_h, mu_h, S_h, hist_h = _iterate(dt, pi0, mu0, S0, 3, False)
    _, _, _, pi_d, mu_d, S_d, hist_d = _iterate(dt, pi0, mu0, S0, 3, True)
    assert np.max(np.abs(hist_d - hist_h) / np.abs(hist_h)) < 1e-14
    scale = lambda a: max(1e-300, np.max(np.abs(a)))
    assert np.max(np.abs(pi_d - pi_h)) / scale(pi_h) < 1e-13
    assert np.max(np.abs(mu_d - mu_h)) / scale(mu_h) < 1e-13
    assert np.max(np.abs(S_d - S_h)) / scale(S_h) < 1e-12
    dt.close()


def test_device_closing_d1024(ctx):
    """The largest dimension of the tier (the E-step's centred tile fills the LDS): one component, two iterations."""
    from ml_amd import _lib
    d, K, n = 1024, 1, 3000
    rng = np.random.default_rng(5)
    X = np.ascontiguousarray(rng.standard_normal((n, d)) * (1.0 + 0.1 * rng.standard_normal(d)))
    dt = _lib.Data(ctx, X)
    mean, cov = dt.sample_covariance()
    S0, pi0, mu0 = cov[None] + 0.5 * np.eye(d)[None], np.ones(1), mean[None] + 0.01
    _, _, _, pi_h, mu_h, S_h, hist_h = _iterate(dt, pi0, mu0, S0, 2, False)
    _, _, _, pi_d, mu_d, S_d, hist_d = _iterate(dt, pi0, mu0, S0, 2, True)
    assert np.max(np.abs(hist_d - hist_h) / np.abs(hist_h)) < 1e-14
    assert np.array_equal(pi_d, pi_h)
    assert np.max(np.abs(mu_d - mu_h)) < 1e-13 * np.max(np.abs(mu_h))
    assert np.max(np.abs(S_d - S_h)) < 1e-12 * np.max(np.abs(S_h))
    dt.close()


def test_flagged_component_goes_through_the_host(ctx, oracle):
    """A far, tight component at d = 80 raises the refinement flag in the device closing: that iteration is closed on the host with
    the refinement pass (as for d <= 64) -- the log-likelihoods follow the oracle's."""
    from ml_amd import _lib
    d, K, n = 80, 2, 6000
    rng = np.random.default_rng(3)
    centres = np.stack([np.zeros(d), np.full(d, 300.0)])
    sig = np.array([1.0, 1e-3])
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(centres[comp] + rng.standard_normal((n, d)) * sig[comp][:, None])
    mu0 = centres + 0.1 * sig[:, None] * rng.standard_normal((K, d))
    S0 = np.stack([np.eye(d) * 1.5 * s * s for s in sig])
    pi0 = np.full(K, 0.5)
    dt = _lib.Data(ctx, X)
    _, _, _, pi, mu, S, hist = _iterate(dt, pi0, mu0, S0, 3, True)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    for it in range(3):
        em.expectation_step(X)
        assert abs(hist[it] - em.log_likelihood) <= 1e-11 * abs(em.log_likelihood)
        em.maximisation_step(X)
    assert np.max(np.abs(pi - em.mixing_probabilities)) < 1e-12
    assert np.max(np.abs(mu - em.means) / np.maximum(1.0, np.abs(em.means))) < 1e-13
    dt.close()
