"""BASELINE.json's full sizes through size-independent properties (the CPU oracle cannot finish these in seconds):
linearity of the statistics in the samples, monotone log-likelihood, normalisation, symmetry / positive-definiteness,
agreement of a full-size run's prefix with the oracle on a sub-block. Needs a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


def test_em_full_size_properties(ctx):
    """N=10M, d=32, K=64 (BASELINE.json configs[2])."""
    from ml_amd import _lib, synth
    n, d, K = 10_000_000, 32, 64
    mix = synth.Mixture(d, K)
    X, comp = mix.sample(n)
    dt = _lib.Data(ctx, X)
    mean, cov = dt.sample_covariance()
    assert np.max(np.abs(mean - X[:200000].mean(axis=0))) < 0.1          # sanity on a prefix
    pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
    lls = []
    for _ in range(6):
        ll, pi, mu, S = dt.em_step(pi, mu, S)
        lls.append(ll)
        assert abs(pi.sum() - 1.0) < 1e-12                               # sum_k S0_k == N
        assert np.array_equal(S, np.transpose(S, (0, 2, 1)))
    assert all(b >= a - 1e-12 * abs(a) for a, b in zip(lls, lls[1:])), lls  # EM never decreases the likelihood
    for k in range(K):
        np.linalg.cholesky(S[k])                                         # positive definite
    # the fit recovers the generating mixture (well separated components, 156k samples each)
    order = np.argsort(mix.weights)
    assert np.max(np.abs(np.sort(pi) - mix.weights[order])) < 2e-3
    nearest = np.argmin(((mu[:, None, :] - mix.means[None, :, :]) ** 2).sum(-1), axis=1)
    assert sorted(nearest.tolist()) == list(range(K))
    assert np.max(np.abs(mu - mix.means[nearest])) < 0.05
    labels = dt.em_labels(K)
    assert np.mean(nearest[labels[:500000]] == comp[:500000]) > 0.999
    # linearity: statistics of the whole block == statistics of its two halves, combined by the closing arithmetic
    ll_full, pi_f, mu_f, S_f = dt.em_step(pi, mu, S)
    dt.close()
    half = n // 2
    outs = []
    for lo, hi in ((0, half), (half, n)):
        part = _lib.Data(ctx, X[lo:hi])
        outs.append((hi - lo,) + part.em_step(pi, mu, S))
        part.close()
    w = np.array([o[0] for o in outs], dtype=float) / n
    ll_comb = sum(wi * o[1] for wi, o in zip(w, outs))
    pi_comb = sum(wi * o[2] for wi, o in zip(w, outs))
    assert abs(ll_comb - ll_full) <= 1e-12 * abs(ll_full)
    assert np.max(np.abs(pi_comb - pi_f)) <= 1e-13
    mu_comb = sum((wi * o[2])[:, None] * o[3] for wi, o in zip(w, outs)) / pi_comb[:, None]
    assert np.max(np.abs(mu_comb - mu_f)) <= 1e-11 * np.max(np.abs(mu_f))


def test_em_prefix_matches_oracle_at_d32_K64(ctx, oracle):
    """Same d, K as the headline configuration on a block the oracle finishes in seconds."""
    from ml_amd import _lib, synth
    d, K, n = 32, 64, 12000
    mix = synth.Mixture(d, K)
    X, _ = mix.sample(n)
    pi0, mu0 = np.full(K, 1.0 / K), mix.initial_means()
    S0 = np.stack([np.cov(X.T)] * K)
    dt = _lib.Data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(pi0, mu0, S0)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    em.calculate_labels()
    assert np.array_equal(dt.em_labels(K), em.labels)
    assert np.max(np.abs(dt.em_responsibilities(K) - em.responsibilities)) < 1e-12
    em.maximisation_step(X)
    assert np.max(np.abs(pi1 - em.mixing_probabilities)) <= 1e-12
    assert np.max(np.abs(mu1 - em.means)) <= 1e-11 * np.max(np.abs(em.means))
    assert np.max(np.abs(S1 - em.covariances)) <= 1e-10 * np.max(np.abs(em.covariances))
    dt.close()


def test_kmeans_large_properties(ctx):
    """K-means at d=8, K=256 (BASELINE.json configs[4] shape) on 20M samples of one GPU's share: per-step invariants."""
    from ml_amd import _lib, synth
    n, d, K = 20_000_000, 8, 256
    mix = synth.Mixture(d, K, seed=77, diagonal=True)
    X, _ = mix.sample(n)
    dt = _lib.Data(ctx, X)
    C = mix.means + 0.3 * np.random.default_rng(1).standard_normal((K, d))
    inertias = []
    for step in range(4):
        inertia, changed, counts, C_new = dt.kmeans_step(C)
        assert counts.sum() == n                                        # every sample assigned exactly once
        assert changed == n if step == 0 else changed < n
        inertias.append(inertia)
        C = C_new
    assert all(b <= a * (1 + 1e-12) for a, b in zip(inertias, inertias[1:])), inertias   # Lloyd never increases inertia
    labels = dt.kmeans_labels()
    assert labels.max() < K
    # the means returned are the means of the assigned samples (checked on a few clusters)
    inertia, changed, counts, C_new = dt.kmeans_step(C)
    labels = dt.kmeans_labels()
    for k in (0, 17, 255):
        sel = X[labels == k]
        assert sel.shape[0] == counts[k]
        assert np.max(np.abs(sel.mean(axis=0) - C_new[k])) <= 1e-11 * max(1.0, np.max(np.abs(C_new[k])))
    # idempotence: assigning again to the same centroids changes nothing
    i2, ch2 = dt.kmeans_assign(C)
    assert ch2 == 0 and abs(i2 - inertia) <= 1e-13 * inertia
    dt.close()
