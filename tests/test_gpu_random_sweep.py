"""Seeded random sweep over shapes (d = 1..128 and a few beyond, K = 1..70, ragged N) comparing one E+M iteration, labels, sample covariance and
one K-means step of the HIP path against the CPU oracle -- exercises every kernel variant (padded dimensions, VALU / MFMA E-step,
narrow / wide statistics kernels, K not a multiple of 16, tiles with a ragged tail). Needs a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


def _cases():
    rng = np.random.default_rng(20241003)
    cases = []
    for d in list(range(1, 33)) + [8, 16, 32, 32] + [33, 36, 40, 41, 47, 48, 50, 56, 57, 63, 64, 64] + [65, 72, 77, 88, 96, 100, 120, 127, 128]:
        K = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 48, 64, 70]))
        n = int(rng.integers(max(8 * K, 70), 2600))
        cases.append((d, K, n, int(rng.integers(1 << 30))))
    # many components; tiny samples (fewer rows than a tile, a wave, or the padded dimension)
    # ... and dimensions beyond 128 (device/generic_dim.hip)
    for d, K, n in ((5, 200, 3000), (16, 130, 2100), (32, 257, 2600), (3, 2, 5), (16, 3, 17), (40, 2, 9), (8, 1, 2), (140, 3, 700), (177, 17, 1500),
                    (256, 2, 600), (131, 1, 300)):
        cases.append((d, K, n, int(rng.integers(1 << 30))))
    return cases


@pytest.fixture(scope="module")
def ctx():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


@pytest.mark.parametrize("d,K,n,seed", _cases())
def test_random_shape(ctx, oracle, d, K, n, seed):
    from ml_amd import _lib
    rng = np.random.default_rng(seed)
    means = 2.0 * rng.standard_normal((K, d)) + rng.uniform(-5, 5, d)      # off-centre data: exercises the shift
    scales = rng.uniform(0.5, 1.5, (K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + scales[comp] * rng.standard_normal((n, d)))
    mu0 = means + 0.3 * rng.standard_normal((K, d))
    S0 = np.empty((K, d, d))
    for k in range(K):
        A = 0.3 * rng.standard_normal((d, d))
        S0[k] = A @ A.T + np.diag(rng.uniform(0.8, 1.6, d))
    pi0 = rng.dirichlet(np.ones(K) * 4)

    dt = _lib.Data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(pi0, mu0, S0)
    R = dt.em_responsibilities(K)
    labels = dt.em_labels(K)
    mean, cov = dt.sample_covariance()
    inertia, changed, counts, C1 = dt.kmeans_step(mu0)
    klabels = dt.kmeans_labels()
    dt.close()

    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * max(1.0, abs(em.log_likelihood))
    assert np.max(np.abs(R - em.responsibilities)) < 1e-12
    em.calculate_labels()
    assert np.array_equal(labels, em.labels)
    em.maximisation_step(X)
    assert relerr(pi1, em.mixing_probabilities) < 1e-11
    assert relerr(mu1, em.means) < 1e-11
    assert relerr(S1, em.covariances) < 1e-9        # components with few samples: cancellation about the global shift
    assert relerr(mean, X.mean(axis=0)) < 1e-13
    assert relerr(cov, oracle.sample_covariance(X)) < 1e-11

    km = oracle.KMeans(K)
    km.set_centroids(mu0, n)
    km.assignment_step(X)
    assert np.array_equal(klabels, km.labels)
    assert abs(inertia - km.inertia) <= 1e-13 * km.inertia
    assert changed == n
    km.update_step(X)
    assert np.array_equal(counts, np.bincount(km.labels, minlength=K).astype(float))
    assert relerr(C1, km.centroids) < 1e-13


def _lattice_cases():
    rng = np.random.default_rng(77)
    cases = []
    for d in (4, 4, 8, 8, 8, 12, 16, 16, 20, 32, 32, 5, 3, 40, 64):
        for K in (int(rng.choice([16, 31, 64, 100])), int(rng.choice([130, 256, 300, 520, 1100, 2500]))):
            cases.append((d, K, int(rng.integers(1500, 4000)), int(rng.integers(1 << 30))))
    return cases


@pytest.mark.parametrize("d,K,n,seed", _lattice_cases())
def test_kmeans_on_an_integer_lattice(ctx, oracle, d, K, n, seed):
    """Samples and centroids on a small integer lattice: distances are small integers, so EXACT ties between clusters
    are everywhere (and every product is exact, so the matrix-core scores tie exactly too). The reference's rule --
    strict '<', first minimum wins (ML/KMeans.cpp:155-163) -- must come out of both tracking modes of the matrix-core
    kernel (per-score for small K, tagged quad maxima for K >= 16 d), of the chunked-table variant and of the VALU
    kernel: labels, counts, inertia and per-sample distances bit for bit."""
    from ml_amd import _lib
    rng = np.random.default_rng(seed)
    C = rng.integers(-3, 4, (K, d)).astype(np.float64)
    X = np.ascontiguousarray(rng.integers(-3, 4, (n, d)).astype(np.float64))
    X[: n // 4] = C[rng.integers(0, K, n // 4)]                      # zero distances as well
    dt = _lib.Data(ctx, X)
    inertia, changed, counts, C1 = dt.kmeans_step(C)
    labels = dt.kmeans_labels()
    dist = dt.min_squared_distances(C)
    dt.close()
    km = oracle.KMeans(K)
    km.set_centroids(C, n)
    km.assignment_step(X)
    assert np.array_equal(labels, km.labels)
    ref = np.array([km.assign_label(X[i])[1] for i in range(0, n, 7)])
    assert np.array_equal(dist[::7], ref)
    assert inertia == km.inertia                                     # sums of small integers: exact in any order
    assert np.array_equal(counts, np.bincount(km.labels, minlength=K).astype(float))
