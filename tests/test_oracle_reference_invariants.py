"""The reference's own C++ test invariants (Tests/test_EM.cpp, Tests/test_KMeans.cpp,
Tests/test_LinearAlgebra.cpp), re-expressed against the CPU restatement in oracle/ on data regenerated
with the same libstdc++ <random> calls. This is what pins the oracle to the reference. CPU only."""
import numpy as np
import pytest


def _check_two_gaussians_em(oracle, init_kind, maximise_first):
    # Tests/test_EM.cpp:8-104
    data, _ = oracle.testdata_two_gaussians(400)
    K, d, n = 2, 3, 400
    em = oracle.EM(K)
    assert not em.converged
    em.set_absolute_tolerance(1e-8)
    em.set_relative_tolerance(1e-8)
    em.set_maximum_steps(100)
    if init_kind is not None:
        em.set_means_initialiser(init_kind)
    em.set_maximise_first(maximise_first)
    em.set_seed(63413131)
    assert em.fit(data), "EM::fit did not converge"
    assert em.converged
    assert em.mixing_probabilities.size == K
    assert em.labels.size == n
    assert em.means.shape == (K, d)
    R = em.responsibilities
    assert R.shape == (n, K)
    for i in range(n):   # :58-62
        u = em.assign_responsibilities(data[i])
        assert np.linalg.norm(u - R[i]) <= 1e-15, i

    means = oracle.TWO_GAUSSIANS_MEANS.copy()
    covs = np.stack([np.diag(s ** 2) for s in oracle.TWO_GAUSSIANS_SIGMAS])
    p0 = oracle.TWO_GAUSSIANS_P0
    probs = np.array([p0, 1 - p0])
    pi = em.mixing_probabilities
    if (pi[0] < pi[1]) != (p0 < 1 - p0):   # :76-81 label swap
        probs = probs[::-1]
        means = means[::-1]
        covs = covs[::-1]
    assert np.linalg.norm(probs - pi) <= 2e-2
    assert np.linalg.norm(means - em.means) <= 2e-2
    for k in range(K):
        assert np.linalg.norm(covs[k] - em.covariances[k]) <= 1e-2

    em1 = oracle.EM(1)   # :89-103
    if init_kind is not None:
        em1.set_means_initialiser(init_kind)
    em1.set_maximise_first(maximise_first)
    em1.fit(data)
    assert em1.log_likelihood <= em.log_likelihood
    assert np.linalg.norm(data.mean(axis=0) - em1.means[0]) <= 1e-14
    R1 = em1.responsibilities
    labels1 = em1.labels
    for i in range(n):
        u = em1.assign_responsibilities(data[i])
        assert np.linalg.norm(u - R1[i]) <= 1e-15
        assert labels1[i] == 0


def test_em_two_gaussians_forgy(oracle): _check_two_gaussians_em(oracle, oracle.FORGY, False)
def test_em_two_gaussians_random_partition(oracle): _check_two_gaussians_em(oracle, oracle.RANDOM_PARTITION, False)
def test_em_two_gaussians_kpp(oracle): _check_two_gaussians_em(oracle, oracle.KPP, False)
def test_em_two_gaussians_closest_mean(oracle): _check_two_gaussians_em(oracle, None, True)


DETERMINISTIC = np.array([[-1, 1, 0.5], [0, 0.5, 0.5]])   # Tests/test_EM.cpp:131-134 as N x d rows


def test_em_deterministic(oracle):
    # Tests/test_EM.cpp:126-144
    em = oracle.EM(2)
    assert em.fit(DETERMINISTIC)
    assert list(em.labels) == [0, 1]
    assert np.array_equal(em.means, DETERMINISTIC)
    assert em.log_likelihood == np.inf


def _check_two_gaussians_kmeans(oracle, init_kind):
    # Tests/test_KMeans.cpp:8-106
    data, truth = oracle.testdata_two_gaussians(400)
    K, d, n = 2, 3, 400
    km = oracle.KMeans(K)
    assert not km.converged
    km.set_absolute_tolerance(1e-8)
    km.set_maximum_steps(100)
    if init_kind is not None:
        km.set_centroids_initialiser(init_kind)
    km.set_seed(63413131)
    assert km.fit(data)
    assert km.converged
    C = km.centroids
    assert C.shape == (K, d)
    labels = km.labels
    inertia = 0.0
    for i in range(n):
        label, dist = km.assign_label(data[i])
        assert label == labels[i]
        assert abs(np.sum((C[label] - data[i]) ** 2) - dist) <= 1e-15
        inertia += dist
    assert abs(inertia - km.inertia) <= 1e-15
    cent = oracle.TWO_GAUSSIANS_MEANS.copy()
    truth = truth.copy()
    if truth[0] != labels[0]:
        truth = 1 - truth
        cent = cent[::-1]
    assert np.linalg.norm(cent - C) <= 2e-2
    assert np.array_equal(truth, labels)

    km.set_seed(63413131)   # :96-99 multi-init
    km.set_number_initialisations(3)
    assert km.fit(data)
    assert km.inertia <= inertia

    km1 = oracle.KMeans(1)
    if init_kind is not None:
        km1.set_centroids_initialiser(init_kind)
    km1.fit(data)
    assert np.linalg.norm(data.mean(axis=0) - km1.centroids[0]) <= 1e-14
    for i in range(n):
        assert km1.assign_label(data[i])[0] == 0


def test_kmeans_two_gaussians_forgy(oracle): _check_two_gaussians_kmeans(oracle, oracle.FORGY)
def test_kmeans_two_gaussians_random_partition(oracle): _check_two_gaussians_kmeans(oracle, oracle.RANDOM_PARTITION)
def test_kmeans_two_gaussians_kpp(oracle): _check_two_gaussians_kmeans(oracle, oracle.KPP)


def test_kmeans_deterministic(oracle):
    # Tests/test_KMeans.cpp:108-128
    km = oracle.KMeans(2)
    assert km.fit(DETERMINISTIC)
    assert km.inertia == 0.0
    assert list(km.labels) == [0, 1]
    assert np.array_equal(km.centroids, DETERMINISTIC)


# ---- Tests/test_LinearAlgebra.cpp ---------------------------------------------------------------------

def test_xAx_symmetric_errors(oracle):
    with pytest.raises(oracle.OracleError) as e:
        oracle.xAx_symmetric(np.zeros((2, 3)), np.zeros(2))
    assert e.value.code == -1
    with pytest.raises(oracle.OracleError) as e:
        oracle.xAx_symmetric(np.zeros((3, 3)), np.zeros(2))
    assert e.value.code == -1


@pytest.mark.parametrize("n", [4, 14, 15, 1024])
def test_xAx_symmetric(oracle, n):
    rng = np.random.default_rng(n)
    A0 = rng.uniform(-1, 1, (n, n))
    A = (A0 + A0.T) / 2
    x = rng.uniform(-1, 1, n)
    expected = x @ A @ x
    assert abs(oracle.xAx_symmetric(A, x) - expected) <= abs(expected) * 1e-14 + 1e-300
    # Only the upper triangle may be read (ML/LinearAlgebra.cpp:18-29).
    Au = np.triu(A) + np.tril(rng.uniform(-1, 1, (n, n)), -1)
    assert oracle.xAx_symmetric(Au, x) == oracle.xAx_symmetric(A, x)


@pytest.mark.parametrize("n", [4, 10, 11, 1024])
def test_xxT(oracle, n):
    x = np.random.default_rng(n).uniform(-1, 1, n)
    expected = np.outer(x, x)
    assert np.linalg.norm(oracle.xxT(x) - expected) <= np.linalg.norm(expected) * 1e-15


@pytest.mark.parametrize("n", [4, 13, 14, 1024])
def test_add_a_xxT(oracle, n):
    rng = np.random.default_rng(n)
    A0 = rng.uniform(-1, 1, (n, n))
    x = rng.uniform(-1, 1, n)
    expected = A0 + 0.6 * np.outer(x, x)
    assert np.linalg.norm(oracle.add_a_xxT(x, A0, 0.6) - expected) <= np.linalg.norm(expected) * 1e-15
    with pytest.raises(oracle.OracleError):
        oracle.add_a_xxT(x, np.zeros((n, n + 1)), 0.6)


def test_bad_arguments(oracle):
    # ML/EM.cpp:35,47,55,63,97,100; ML/KMeans.cpp:21,56,59,124,132,140
    with pytest.raises(oracle.OracleError) as e:
        oracle.EM(0)
    assert e.value.code == -1
    em = oracle.EM(2)
    with pytest.raises(oracle.OracleError) as e:
        em.set_absolute_tolerance(-1.0)
    assert e.value.code == -2
    with pytest.raises(oracle.OracleError) as e:
        em.set_relative_tolerance(-1.0)
    assert e.value.code == -2
    with pytest.raises(oracle.OracleError) as e:
        em.set_maximum_steps(1)
    assert e.value.code == -1
    with pytest.raises(oracle.OracleError) as e:
        em.fit(np.zeros((1, 3)))
    assert e.value.code == -1
    with pytest.raises(oracle.OracleError):
        oracle.KMeans(0)
    km = oracle.KMeans(2)
    with pytest.raises(oracle.OracleError) as e:
        km.set_absolute_tolerance(-1.0)
    assert e.value.code == -2
    with pytest.raises(oracle.OracleError):
        km.set_maximum_steps(1)
    with pytest.raises(oracle.OracleError):
        km.set_number_initialisations(0)
    with pytest.raises(oracle.OracleError):
        km.fit(np.zeros((1, 3)))
