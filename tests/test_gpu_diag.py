"""Diagonal-covariance EM (BASELINE.json configs[1]; an extension -- the reference is full-covariance only, ML/EM.hpp:175):
the fused HIP kernel (device/em_diag.hip) through the C ABI against the scikit-learn covariance_type='diag' fixtures and the
oracle's diagonal mode. Tolerances as for the full path: log-likelihood 1e-12 relative, mixing/means 1e-11, variances
1e-10 (max-norm relative), responsibilities 1e-12 absolute, labels bit-exact."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

DIAG_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "em_onestep_diag_*.npz")))


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def ctx():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


def _data(ctx, X):
    from ml_amd import _lib
    return _lib.Data(ctx, np.ascontiguousarray(X, dtype=np.float64))


def _oracle_step(oracle, X, pi0, mu0, var0):
    K = pi0.size
    em = oracle.EM(K)
    em.set_covariance_type("diag")
    em.set_parameters(mu0, np.stack([np.diag(v) for v in var0]), pi0)
    em.expectation_step(X)
    ll, R = em.log_likelihood, em.responsibilities.copy()
    em.calculate_labels()
    labels = em.labels.copy()
    em.maximisation_step(X)
    S = em.covariances
    return ll, R, labels, em.mixing_probabilities.copy(), em.means.copy(), np.stack([np.diag(S[k]) for k in range(K)])


@pytest.mark.parametrize("case", DIAG_CASES)
def test_diag_step_matches_sklearn_fixture(ctx, case):
    g = load_golden(case)
    X = g["X"]
    K = g["pi0"].size
    dt = _data(ctx, X)
    ll, pi1, mu1, var1 = dt.em_step_diag(g["pi0"], g["mu0"], g["var0"])
    assert abs(ll - float(g["ll0"])) <= 1e-12 * abs(float(g["ll0"]))
    assert relerr(pi1, g["pi1"]) < 1e-11
    assert relerr(mu1, g["mu1"]) < 1e-11
    assert relerr(var1, g["var1"]) < 1e-10
    # the N x K block is rebuilt on demand from the same parameters
    assert np.max(np.abs(dt.em_responsibilities(K) - g["R0"])) < 1e-12
    assert np.array_equal(dt.em_labels(K), g["labels0"])
    # a K x d x d stack of diagonal matrices is accepted and returned in kind
    ll_b, pi_b, mu_b, S_b = dt.em_step_diag(g["pi0"], g["mu0"], np.stack([np.diag(v) for v in g["var0"]]))
    assert ll_b == ll and np.array_equal(pi_b, pi1) and np.array_equal(mu_b, mu1)
    assert np.array_equal(S_b, np.stack([np.diag(v) for v in var1]))
    dt.close()


@pytest.mark.parametrize("n,d,K", [(1000, 1, 1), (777, 2, 3), (5000, 3, 16), (4099, 5, 17), (3000, 7, 33), (2500, 8, 64),
                                   (6000, 12, 5), (20000, 16, 16), (3001, 20, 48), (2000, 24, 2), (2000, 28, 31),
                                   (4000, 32, 64), (63, 4, 2), (64, 6, 4), (65, 16, 16), (3000, 8, 9), (300, 16, 1), (129, 6, 16)])
def test_diag_step_matches_oracle(ctx, oracle, n, d, K):
    rng = np.random.default_rng(1000 * d + K)
    means = 3.0 * rng.standard_normal((K, d))
    var = rng.uniform(0.3, 2.0, (K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)) * np.sqrt(var[comp]))
    mu0 = means + 0.2 * rng.standard_normal((K, d))
    var0 = var * rng.uniform(0.8, 1.25, (K, d))
    pi0 = rng.dirichlet(np.ones(K) * 4)
    ll0, R0, labels0, pi_o, mu_o, var_o = _oracle_step(oracle, X, pi0, mu0, var0)
    dt = _data(ctx, X)
    ll, pi1, mu1, var1 = dt.em_step_diag(pi0, mu0, var0)
    assert abs(ll - ll0) <= 1e-12 * abs(ll0)
    assert relerr(pi1, pi_o) < 1e-11 and relerr(mu1, mu_o) < 1e-11 and relerr(var1, var_o) < 1e-10
    assert np.max(np.abs(dt.em_responsibilities(K) - R0)) < 1e-12
    srt = np.sort(R0, axis=1)
    clear = (srt[:, -1] - srt[:, -2] > 1e-9) if K > 1 else np.ones(n, bool)     # (exact ties aside, none expected)
    assert np.array_equal(dt.em_labels(K)[clear], labels0[clear])
    # several iterations stay on the oracle's trajectory
    em_pi, em_mu, em_var = pi1, mu1, var1
    o_pi, o_mu, o_var = pi_o, mu_o, var_o
    for _ in range(3):
        ll_g, em_pi, em_mu, em_var = dt.em_step_diag(em_pi, em_mu, em_var)
        ll_o, _, _, o_pi, o_mu, o_var = _oracle_step(oracle, X, o_pi, o_mu, o_var)
        assert abs(ll_g - ll_o) <= 1e-11 * abs(ll_o)
    assert relerr(em_mu, o_mu) < 1e-9 and relerr(em_var, o_var) < 1e-9
    dt.close()


def test_tight_clusters_far_from_the_global_mean_are_refined(ctx, oracle):
    """Variances of components far (in their own sigmas) from the shared statistics shift come from the second pass."""
    rng = np.random.default_rng(5)
    d, K, n = 6, 3, 30000
    centres = np.array([[0.0] * d, [1000.0] * d, [-500.0] * d])
    sig = np.array([1.0, 1e-3, 1e-2])
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(centres[comp] + rng.standard_normal((n, d)) * sig[comp][:, None])
    mu0 = centres + 0.1 * sig[:, None] * rng.standard_normal((K, d))
    var0 = np.repeat((sig ** 2)[:, None], d, axis=1) * 1.5
    pi0 = np.full(K, 1.0 / K)
    _, _, _, pi_o, mu_o, var_o = _oracle_step(oracle, X, pi0, mu0, var0)
    dt = _data(ctx, X)
    _, pi1, mu1, var1 = dt.em_step_diag(pi0, mu0, var0)
    assert relerr(pi1, pi_o) < 1e-12
    assert np.max(np.abs(mu1 - mu_o) / np.maximum(1.0, np.abs(mu_o))) < 1e-13
    assert np.max(np.abs(var1 - var_o) / var_o) < 1e-9
    dt.close()


@pytest.mark.parametrize("d,K,n", [(40, 3, 3000), (4, 65, 6000), (100, 5, 4000), (24, 130, 9000)])
def test_shapes_beyond_the_fused_kernel_run_through_the_full_covariance_kernels(ctx, oracle, d, K, n):
    """d > 32 or K > 64: the diagonal mode is not refused any more (missing #5) -- the iteration goes through the
    full-covariance kernels on diagonal matrices and must agree with the oracle's diagonal mode like the fused kernel does."""
    rng = np.random.default_rng(d * 1000 + K)
    centres = 4.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(centres[rng.integers(0, K, n)] + rng.standard_normal((n, d)) * rng.uniform(0.5, 1.5, d))
    mu0 = centres + 0.2 * rng.standard_normal((K, d))
    var0 = np.tile(np.var(X, axis=0), (K, 1)) * rng.uniform(0.8, 1.2, (K, d))
    pi0 = rng.dirichlet(np.full(K, 5.0))
    dt = _data(ctx, X)
    ll, pi1, mu1, var1 = dt.em_step_diag(pi0, mu0, var0)
    em = oracle.EM(K)
    em.set_covariance_type("diag")
    em.set_parameters(mu0, np.stack([np.diag(v) for v in var0]), pi0)
    em.expectation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    em.calculate_labels()
    assert np.array_equal(dt.em_labels(K), np.asarray(em.labels))
    em.maximisation_step(X)
    var_o = np.stack([np.diag(c) for c in em.covariances])
    assert relerr(pi1, em.mixing_probabilities) < 1e-11 and relerr(mu1, em.means) < 1e-11
    assert np.max(np.abs(var1 - var_o) / var_o) < 1e-9
    steps, conv, ll_it, pi_b, mu_b, var_b, hist = dt.em_iterate(pi0, mu0, var0, 3, 0.0, 0.0, True)
    assert steps == 3 and abs(hist[0] - ll) <= 1e-13 * abs(ll)
    dt.close()


def test_facade_fit_in_diagonal_mode_matches_the_oracle_fit(oracle):
    """cppyml.clustering.EM.set_covariance_type('diag'): whole fits (both start modes) against the oracle's diagonal fits --
    same number of steps, log-likelihood 1e-12, parameters 1e-10, labels bit-exact."""
    from ml_amd.cppyml import clustering as cl
    g = load_golden("em_onestep_diag_d16_K16.npz")
    X = g["X"]
    K, d = g["mu0"].shape
    for maximise_first in (False, True):
        em = cl.EM(K)
        em.set_covariance_type("diag")
        em.set_means_initialiser(cl.FixedCentroids(g["mu0"]))
        em.set_responsibilities_initialiser(cl.ClosestCentroid(cl.FixedCentroids(g["mu0"])))
        em.set_maximise_first(maximise_first)
        em.set_absolute_tolerance(1e-11)
        em.set_relative_tolerance(1e-11)
        em.set_maximum_steps(300)
        ref = oracle.EM(K)
        ref.set_covariance_type("diag")
        ref.set_means_initialiser(oracle.FIXED, g["mu0"])
        ref.set_responsibilities_initialiser(oracle.FIXED, g["mu0"])
        ref.set_maximise_first(maximise_first)
        ref.set_absolute_tolerance(1e-11)
        ref.set_relative_tolerance(1e-11)
        ref.set_maximum_steps(300)
        assert em.fit(X) and ref.fit(X)
        assert em.steps_done == ref.steps_done
        assert abs(em.log_likelihood - ref.log_likelihood) <= 1e-12 * abs(ref.log_likelihood)
        assert relerr(em.means.T, ref.means) < 1e-10
        assert relerr(em.mixing_probabilities, ref.mixing_probabilities) < 1e-10
        S = ref.covariances
        for k in range(K):
            C = em.covariance(k)
            assert np.all(C - np.diag(np.diag(C)) == 0)
            assert relerr(np.diag(C), np.diag(S[k])) < 1e-10
        assert np.array_equal(np.asarray(em.labels), ref.labels)
        u = em.assign_responsibilities(X[0])
        assert np.max(np.abs(u - em.responsibilities[0])) < 1e-13


def test_config_b_full_size_properties(ctx):
    """BASELINE.json configs[1] at its full size, N=1M, d=16, K=16 (the oracle would need minutes): identities every exact
    M-step satisfies, checked against numpy column moments -- sum_k pi_k = 1, sum_k pi_k mu_k = mean(X),
    sum_k pi_k (var_k + mu_k^2) = mean(X^2) -- plus a non-decreasing log-likelihood and agreement with the oracle-checked
    small-N path on a prefix of the same rows."""
    from ml_amd import synth
    n, d, K = 1_000_000, 16, 16
    mix = synth.Mixture(d, K, diagonal=True)
    X, _ = mix.sample(n)
    dt = _data(ctx, X)
    _, cov = dt.sample_covariance()
    pi, mu, var = np.full(K, 1.0 / K), mix.initial_means(), np.repeat(np.diag(cov)[None, :], K, axis=0)
    m1, m2 = X.mean(axis=0), (X * X).mean(axis=0)
    last = -np.inf
    for it in range(6):
        ll, pi, mu, var = dt.em_step_diag(pi, mu, var)
        assert ll >= last - 1e-12 * abs(ll)
        last = ll
        assert abs(pi.sum() - 1.0) < 1e-12
        assert np.max(np.abs(pi @ mu - m1)) < 1e-11 * np.max(np.abs(m1)) + 1e-12
        assert np.max(np.abs(pi @ (var + mu * mu) - m2) / m2) < 1e-11
    labels = dt.em_labels(K)
    assert labels.shape == (n,) and labels.max() < K
    dt.close()


@pytest.mark.parametrize("d,K,n", [(16, 16, 6000), (5, 17, 3000), (8, 40, 5000), (32, 64, 4000)])
def test_two_operation_density_form_and_its_guard(ctx, oracle, d, K, n, monkeypatch):
    """The diagonal kernel evaluates ((x - mu) / sigma)^2 as fma(a, x~, b)^2 on shift-centred coordinates while every
    |b| = |mu - shift| / sigma is below kDiagAbLimit, and in the exact form (x - mu first) otherwise. Ordinary data: both forms
    agree to rounding (and with the oracle within the usual tolerances). A tight component far from the global mean: the
    guard selects the exact form by itself -- bit-identical to a run with the two-operation form switched off."""
    rng = np.random.default_rng(100 * d + K)
    means = 3.0 * rng.standard_normal((K, d))
    sig = rng.uniform(0.6, 1.5, (K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + sig[comp] * rng.standard_normal((n, d)))
    pi0 = np.full(K, 1.0 / K)
    mu0 = means + 0.1 * rng.standard_normal((K, d))
    var0 = sig ** 2

    def step(Xd, mu, var, off):
        if off:
            monkeypatch.setenv("MLHIP_DIAG_AB", "0")
        else:
            monkeypatch.delenv("MLHIP_DIAG_AB", raising=False)
        dt = _data(ctx, Xd)
        out = dt.em_step_diag(pi0, mu, var)
        dt.close()
        monkeypatch.delenv("MLHIP_DIAG_AB", raising=False)
        return out

    a, b = step(X, mu0, var0, False), step(X, mu0, var0, True)
    assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0])
    assert relerr(a[1], b[1]) < 1e-12 and relerr(a[2], b[2]) < 1e-12 and relerr(a[3], b[3]) < 1e-11
    ll0, _, _, pi_o, mu_o, var_o = _oracle_step(oracle, X, pi0, mu0, var0)
    assert abs(a[0] - ll0) <= 1e-12 * abs(ll0)
    assert relerr(a[1], pi_o) < 1e-11 and relerr(a[2], mu_o) < 1e-11 and relerr(a[3], var_o) < 1e-10

    # component 0 becomes tight and far: |mu - shift| / sigma ~ 1e5 >> the limit
    X2 = X.copy()
    far = 50.0 + np.arange(d)
    X2[comp == 0] = far + 1e-3 * rng.standard_normal((int(np.sum(comp == 0)), d))
    mu2, var2 = mu0.copy(), var0.copy()
    mu2[0], var2[0] = far, 1e-6
    c, e = step(X2, mu2, var2, False), step(X2, mu2, var2, True)
    assert c[0] == e[0] and np.array_equal(c[1], e[1]) and np.array_equal(c[2], e[2]) and np.array_equal(c[3], e[3])


def _experiments_build():
    from ml_amd import _lib
    return os.path.basename(_lib.LIB_PATH) == "libmlhip_exp.so"


@pytest.mark.skipif(not _experiments_build(), reason="em_diag_sgpr_kernel is an A/B variant: `make -C ml_amd/csrc EXPERIMENTS=1`, MLHIP_LIBRARY=.../libmlhip_exp.so")
@pytest.mark.parametrize("d,K,n", [(16, 16, 20000), (12, 7, 9000), (6, 16, 9000), (3, 4, 5000)])
def test_matrix_core_density_form_and_its_guard(ctx, oracle, d, K, n, monkeypatch):
    """MLHIP_DIAG_GEMM=1 (an A/B variant, in `make EXPERIMENTS=1` builds only): the log-densities as ONE product [K x 2d] . [x~^2 ; x~] on the matrix
    cores -- the expanded form, whose cancellation costs about 4 eps B2 in a log-responsibility, B2_k = |(mu_k - shift) / sigma_k|^2.
    Overlapping components whose PAIRS sit far from the global mean (B2 of several hundred, responsibilities strictly between 0
    and 1): within the usual tolerances of the oracle while every B2 is below the limit; beyond it the kernel takes the exact
    scalar-fed form by itself -- bit-identical to the run without the variant."""
    rng = np.random.default_rng(7 * d + K)
    centres = 16.0 * rng.standard_normal((K // 2 + 1, d)) / np.sqrt(d)          # |centre| ~ 16 sigma: B2 of a few hundred
    means = centres[np.arange(K) // 2] + 0.6 * rng.standard_normal((K, d)) / np.sqrt(d)   # the two of a pair overlap
    sig = rng.uniform(0.8, 1.25, (K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + sig[comp] * rng.standard_normal((n, d)))
    pi0 = np.full(K, 1.0 / K)
    mu0 = means + 0.05 * rng.standard_normal((K, d))
    var0 = sig ** 2
    b2 = np.max(np.sum((mu0 - X.mean(axis=0)) ** 2 / var0, axis=1))
    assert 50.0 < b2 < 1000.0, b2

    def step(gemm, limit=None):
        monkeypatch.setenv("MLHIP_DIAG_GEMM", "1" if gemm else "0")
        monkeypatch.setenv("MLHIP_DIAG_MIXED", "0")                               # (the scalar-fed kernel hosts the variant)
        if limit is not None:
            monkeypatch.setenv("MLHIP_DIAG_EXPAND_LIMIT", str(limit))
        dt = _data(ctx, X)
        out = dt.em_step_diag(pi0, mu0, var0)
        R, labels = dt.em_responsibilities(K), dt.em_labels(K)
        dt.close()
        for name in ("MLHIP_DIAG_GEMM", "MLHIP_DIAG_MIXED", "MLHIP_DIAG_EXPAND_LIMIT"):
            monkeypatch.delenv(name, raising=False)
        return out, R, labels

    ll0, R0, labels0, pi_o, mu_o, var_o = _oracle_step(oracle, X, pi0, mu0, var0)
    assert np.mean((R0.max(axis=1) < 0.99)) > 0.2                                  # genuinely overlapping
    (ll, pi1, mu1, var1), R, labels = step(True)
    assert abs(ll - ll0) <= 1e-12 * abs(ll0)
    assert relerr(pi1, pi_o) < 1e-11 and relerr(mu1, mu_o) < 1e-11 and relerr(var1, var_o) < 1e-10
    assert np.max(np.abs(R - R0)) < 1e-12                                          # (the block is rebuilt by the exact E-step kernel)
    exact = step(False)
    assert relerr(pi1, exact[0][1]) < 1e-12 and relerr(var1, exact[0][3]) < 1e-11  # ... so the statistics are what shows the form
    assert ll != exact[0][0] or not np.array_equal(mu1, exact[0][2])               # the variant did run
    guarded = step(True, limit=0.5 * b2)                                            # some B2 above the limit: exact form
    assert guarded[0][0] == exact[0][0]
    for a, b in zip(guarded[0][1:], exact[0][1:]):
        assert np.array_equal(a, b)
