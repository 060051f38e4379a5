"""Dimensions beyond 128: the reference has no dimension limit (ML/EM.cpp:96-101). Up to d = 1024 the E-step and the statistics run
on the matrix cores as plain matrix products (ml_amd/csrc/device/big_dim.hip), above that -- and with MLHIP_BIG_DIM=0 -- in a plain
form (generic_dim.hip). One E + M iteration, labels, sample covariance, one K-means step, the step loops and a facade fit against
the oracle at d = 129 … 600 (the reference's own E-step underflows around d = 1000: exp(-q / 2) with q ~ d; the library
works in the log domain -- there the two tiers are held against each other) (K not a multiple of 16, ragged N), both tiers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


def _problem(d, K, n, seed):
    rng = np.random.default_rng(seed)
    means = 2.0 * rng.standard_normal((K, d)) + rng.uniform(-3, 3, d)
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.uniform(0.5, 1.5, (K, d))[comp] * rng.standard_normal((n, d)))
    mu0 = means + 0.2 * rng.standard_normal((K, d))
    S0 = np.empty((K, d, d))
    for k in range(K):
        A = 0.1 * rng.standard_normal((d, d))
        S0[k] = A @ A.T + np.diag(rng.uniform(0.8, 1.6, d))
    pi0 = rng.dirichlet(np.ones(K) * 4)
    return X, pi0, mu0, S0


@pytest.mark.parametrize("tier", ["matrix-core", "plain"])
@pytest.mark.parametrize("d,K,n", [(129, 3, 1500), (160, 5, 2100), (200, 2, 1111), (256, 4, 1800), (333, 2, 900), (191, 19, 4001), (192, 6, 1500), (320, 2, 1000),
                                   (512, 3, 1300), (600, 2, 700)])
def test_one_iteration_labels_covariance_and_kmeans_step_match_the_oracle(oracle, d, K, n, tier, monkeypatch):
    from ml_amd import _lib
    if tier == "plain":
        monkeypatch.setenv("MLHIP_BIG_DIM", "0")
    X, pi0, mu0, S0 = _problem(d, K, n, 7 * d + K)
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    assert not dt.em_plan(K)["matrix_estep"] and not dt.em_plan(K)["self_norm"]
    ll, pi1, mu1, S1 = dt.em_step(pi0, mu0, S0)
    R = dt.em_responsibilities(K)
    labels = dt.em_labels(K)
    mean, cov = dt.sample_covariance()
    inertia, changed, counts, C1 = dt.kmeans_step(mu0)
    klabels, kdist = dt.kmeans_labels(), dt.kmeans_distances()
    dt.close()
    ctx.close()

    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    assert np.max(np.abs(R - em.responsibilities)) < 1e-11
    em.calculate_labels()
    assert np.array_equal(labels, em.labels)
    em.maximisation_step(X)
    assert relerr(pi1, em.mixing_probabilities) < 1e-11 and relerr(mu1, em.means) < 1e-11 and relerr(S1, em.covariances) < 1e-9
    assert relerr(mean, X.mean(axis=0)) < 1e-13 and relerr(cov, oracle.sample_covariance(X)) < 1e-11

    km = oracle.KMeans(K)
    km.set_centroids(mu0, n)
    km.assignment_step(X)
    assert np.array_equal(klabels, km.labels)
    assert abs(inertia - km.inertia) <= 1e-13 * km.inertia and changed == n
    ref = np.array([km.assign_label(X[i])[1] for i in range(0, n, 37)])
    assert np.array_equal(kdist[::37], ref)                       # distances bit-identical to the oracle's fma chain
    km.update_step(X)
    assert np.array_equal(counts, np.bincount(km.labels, minlength=K).astype(float)) and relerr(C1, km.centroids) < 1e-13


def test_fits_through_the_step_loops_and_the_facade_at_d160(oracle):
    from ml_amd import _lib
    from ml_amd.cppyml import clustering as cl
    d, K, n = 160, 3, 2400
    X, pi0, mu0, S0 = _problem(d, K, n, 99)
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    steps, conv, ll, pi, mu, S, hist = dt.em_iterate(np.full(K, 1.0 / K), mu0, np.stack([cov] * K), 6, 0.0, 0.0)
    em = oracle.EM(K)
    em.set_parameters(mu0, np.stack([np.ascontiguousarray(oracle.sample_covariance(X))] * K), np.full(K, 1.0 / K))
    lls = []
    for _ in range(6):
        em.expectation_step(X)
        em.maximisation_step(X)
        lls.append(em.log_likelihood)
    assert steps == 6 and np.max(np.abs(hist - np.array(lls)) / np.abs(np.array(lls))) <= 1e-11
    assert relerr(mu, em.means) < 1e-9 and relerr(S, em.covariances) < 1e-8
    ks, kconv, inertia, counts, C, _ = dt.kmeans_iterate(mu0, 100, 0.0)
    km = oracle.KMeans(K)
    km.set_absolute_tolerance(0.0)
    km.set_maximum_steps(100)
    km.set_centroids_initialiser(oracle.FIXED, mu0)
    assert km.fit(X) == kconv and km.steps_done == ks
    assert np.array_equal(km.labels, dt.kmeans_labels()) and relerr(C, km.centroids) < 1e-12
    dt.close()
    ctx.close()

    fit = cl.EM(K)
    fit.set_means_initialiser(cl.FixedCentroids(mu0))
    fit.set_maximum_steps(6)
    fit.set_absolute_tolerance(0.0)
    fit.set_relative_tolerance(0.0)
    fit.fit(X)
    assert fit.steps_done == 6 and abs(fit.log_likelihood - lls[-1]) <= 1e-11 * abs(lls[-1])


def test_diagonal_mode_initialisers_and_seeded_kmeans_fit_at_d150(oracle):
    """The callers around the three passes at d > 128: the diagonal mode (full-covariance kernels on diagonal matrices), K-means++ and
    RandomPartition with their distance passes / running means on the device, a seeded KMeans.fit -- against the oracle."""
    from ml_amd import _lib
    from ml_amd.cppyml import clustering as cl
    d, K, n = 150, 4, 2000
    X, pi0, mu0, _ = _problem(d, K, n, 5)
    rng = np.random.default_rng(3)
    var0 = np.tile(np.var(X, axis=0), (K, 1)) * rng.uniform(0.8, 1.2, (K, d))
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    ll, pi1, mu1, var1 = dt.em_step_diag(pi0, mu0, var0)
    em = oracle.EM(K)
    em.set_covariance_type("diag")
    em.set_parameters(mu0, np.stack([np.diag(v) for v in var0]), pi0)
    em.expectation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    em.maximisation_step(X)
    var_o = np.stack([np.diag(c) for c in em.covariances])
    assert relerr(mu1, em.means) < 1e-11 and np.max(np.abs(var1 - var_o) / var_o) < 1e-9
    dt.close()
    ctx.close()

    for kind_cl, kind_or in ((cl.KPP(), oracle.KPP), (cl.RandomPartition(), oracle.RANDOM_PARTITION), (cl.Forgy(), oracle.FORGY)):
        km = cl.KMeans(K)
        km.set_centroids_initialiser(kind_cl)
        km.set_seed(11)
        km.set_maximum_steps(5)
        km.fit(X)
        okm = oracle.KMeans(K)
        okm.set_centroids_initialiser(kind_or)
        okm.set_seed(11)
        okm.set_maximum_steps(5)
        okm.fit(X)
        assert np.array_equal(np.array(km.labels), okm.labels)
        assert np.max(np.abs(km.centroids - okm.centroids)) <= 1e-13 * np.max(np.abs(okm.centroids))


@pytest.mark.parametrize("d,K,n", [(130, 40, 3000), (256, 70, 5000), (300, 8, 2000), (512, 33, 1500), (200, 7, 4097), (136, 300, 9000), (900, 20, 1200), (1500, 12, 800)])
def test_kmeans_iterations_agree_between_the_tiers_bit_for_bit(d, K, n, monkeypatch):
    """K-means at d > 128: the register-blocked assignment kernel (big_dim.hip) evaluates the reference's own fma chain, so
    labels, distances, counts and centroids of a step loop equal the plain tier's (MLHIP_BIG_DIM=0) bit for bit -- also for
    few clusters, where the dimension-major table does not fit its scratch and the plain kernel runs either way."""
    from ml_amd import _lib
    rng = np.random.default_rng(d + K + n)
    C = 2.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(C[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    C0 = C + 0.3 * rng.standard_normal((K, d))
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    got = dt.kmeans_iterate(C0, 6, 0.0)
    lg, dg = dt.kmeans_labels(), dt.kmeans_distances()
    monkeypatch.setenv("MLHIP_BIG_DIM", "0")
    ref = dt.kmeans_iterate(C0, 6, 0.0)
    assert got[0] == ref[0] and got[1] == ref[1]
    assert abs(got[2] - ref[2]) <= 1e-14 * ref[2]          # (the inertia: the same distances summed over differently shaped workgroups)
    for a, b in zip(got[3:], ref[3:]):
        assert np.array_equal(a, b)
    assert np.array_equal(lg, dt.kmeans_labels()) and np.array_equal(dg, dt.kmeans_distances())
    dt.close()
    ctx.close()


@pytest.mark.parametrize("d,K,n", [(700, 3, 900), (1024, 2, 500), (1000, 5, 1500)])
def test_matrix_core_tier_against_the_plain_tier_beyond_the_reference_range(d, K, n, monkeypatch):
    """d ~ 1000: the reference's E-step (and with it the oracle's) underflows -- exp(-q / 2), q ~ d -- while the library's log-domain
    passes do not; the matrix-core tier (d <= 1024) is held against the plain tier there: log-likelihood, responsibilities, labels,
    one M-step."""
    from ml_amd import _lib
    X, pi0, mu0, S0 = _problem(d, K, n, 11 * d + K)
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    got = dt.em_step(pi0, mu0, S0)
    R, labels = dt.em_responsibilities(K), dt.em_labels(K)
    monkeypatch.setenv("MLHIP_BIG_DIM", "0")
    ref = dt.em_step(pi0, mu0, S0)
    assert np.isfinite(ref[0]) and abs(got[0] - ref[0]) <= 1e-12 * abs(ref[0])
    assert np.max(np.abs(R - dt.em_responsibilities(K))) < 1e-11 and np.array_equal(labels, dt.em_labels(K))
    for a, b, tol in zip(got[1:], ref[1:], (1e-11, 1e-11, 1e-9)):
        assert relerr(a, b) < tol
    dt.close()
    ctx.close()
