"""Host-side logic of the product (no GPU needed): argument validation and exceptions, exact-fit paths, the
LinearAlgebra helpers, the libstdc++-RNG initialisers (bit-identical to the CPU oracle), the covariance processing
and the M-step closing arithmetic. Mirrors Tests/test_LinearAlgebra.cpp and the error paths of ML/EM.cpp, ML/KMeans.cpp."""
import ctypes as C

import numpy as np
import pytest

from ml_amd import _lib
from ml_amd.cppyml import clustering

lib = _lib.lib
dp = _lib.dptr


def test_constructor_and_setter_errors():
    # ML/EM.cpp:35,47,55,63,71,79 and ML/KMeans.cpp:21,124,132,140,148; pybind11 maps both exception types to ValueError
    with pytest.raises(ValueError, match="At least one component"):
        clustering.EM(0)
    em = clustering.EM(3)
    with pytest.raises(ValueError, match="Negative absolute tolerance"):
        em.set_absolute_tolerance(-1e-3)
    with pytest.raises(ValueError, match="Negative relative tolerance"):
        em.set_relative_tolerance(-1e-3)
    with pytest.raises(ValueError, match="At least two steps"):
        em.set_maximum_steps(1)
    with pytest.raises(ValueError, match="Null means initialiser"):
        em.set_means_initialiser(None)
    with pytest.raises(ValueError, match="Null responsibilities initialiser"):
        em.set_responsibilities_initialiser(None)
    with pytest.raises(ValueError, match="Null centroids initialiser"):
        clustering.ClosestCentroid(None)
    with pytest.raises(ValueError, match="Bad component index"):
        em.fit(np.eye(3)) and em.covariance(3)
    with pytest.raises(ValueError, match="cannot be zero"):
        clustering.KMeans(0)
    km = clustering.KMeans(2)
    with pytest.raises(ValueError, match="Negative absolute tolerance"):
        km.set_absolute_tolerance(-1.0)
    with pytest.raises(ValueError, match="At least two steps"):
        km.set_maximum_steps(1)
    with pytest.raises(ValueError, match="At least 1 initialisation"):
        km.set_number_initialisations(0)
    with pytest.raises(ValueError, match="Null centroids initialiser"):
        km.set_centroids_initialiser(None)
    # valid settings are accepted
    em.set_absolute_tolerance(0.0); em.set_relative_tolerance(0.0); em.set_maximum_steps(2)
    em.set_seed(5); em.set_verbose(False); em.set_maximise_first(True)
    em.set_means_initialiser(clustering.KPP())
    em.set_responsibilities_initialiser(clustering.ClosestCentroid(clustering.RandomPartition()))
    km.set_number_initialisations(centroids_initialiser=3)   # the reference's keyword really is mis-named (clustering.cpp:159)


def test_fit_argument_errors():
    em = clustering.EM(3)
    with pytest.raises(ValueError, match="Not enough data"):
        em.fit(np.zeros((2, 4)))
    with pytest.raises(ValueError, match="At least one dimension"):
        em.fit(np.zeros((5, 0)))
    with pytest.raises(TypeError):
        em.fit(np.zeros((5, 3), dtype=np.float32))          # noconvert(): no silent casting
    with pytest.raises(TypeError):
        em.fit(np.asfortranarray(np.zeros((5, 3))))
    with pytest.raises(TypeError):
        em.fit([[0.0, 1.0]] * 5)
    km = clustering.KMeans(3)
    with pytest.raises(ValueError, match="Not enough data"):
        km.fit(np.zeros((2, 4)))
    with pytest.raises(ValueError, match="At least one dimension"):
        km.fit(np.zeros((5, 0)))
    with pytest.raises(TypeError):
        km.fit(np.zeros((5, 3), dtype=np.int64))


def test_abstract_initialisers_have_no_constructor():
    with pytest.raises(TypeError, match="No constructor defined"):
        clustering.CentroidsInitialiser()
    with pytest.raises(TypeError, match="No constructor defined"):
        clustering.ResponsibilitiesInitialiser()
    assert issubclass(clustering.Forgy, clustering.CentroidsInitialiser)
    assert issubclass(clustering.ClosestCentroid, clustering.ResponsibilitiesInitialiser)


DETERMINISTIC = np.array([[-1, 1, 0.5], [0, 0.5, 0.5]])   # Tests/test_EM.cpp:131-134, one sample per row


def test_em_deterministic_exact_fit():
    # Tests/test_EM.cpp:126-144 -- N == K needs no GPU
    em = clustering.EM(2)
    assert em.fit(DETERMINISTIC)
    assert em.converged
    assert list(em.labels) == [0, 1]
    assert np.array_equal(em.means, DETERMINISTIC.T)        # d x K
    assert em.log_likelihood == np.inf
    assert np.array_equal(em.responsibilities, np.eye(2))
    assert np.array_equal(em.covariance(0), np.zeros((3, 3)))
    assert np.array_equal(em.mixing_probabilities, [0.5, 0.5])


def test_kmeans_deterministic_exact_fit():
    # Tests/test_KMeans.cpp:108-128
    km = clustering.KMeans(2)
    assert km.fit(DETERMINISTIC)
    assert km.inertia == 0.0
    assert km.labels == [0, 1]
    assert np.array_equal(km.centroids, DETERMINISTIC)      # K x d
    for i in range(2):
        assert km.assign_label(DETERMINISTIC[i]) == (i, 0.0)
    km.set_number_initialisations(3)
    assert km.fit(DETERMINISTIC) and km.inertia == 0.0


# ---- LinearAlgebra (Tests/test_LinearAlgebra.cpp) ---------------------------------------------------------------

def _xAx(A, x):
    A = np.asfortranarray(A, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = C.c_double()
    _lib.check(lib.mlpp_xAx_symmetric(dp(A), A.shape[0], A.shape[1], dp(x), x.size, C.byref(out)))
    return out.value


def _xxT(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty((x.size, x.size), order="F")
    _lib.check(lib.mlpp_xxT(dp(x), x.size, dp(out)))
    return out


def _add_a_xxT(x, dest, a):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.array(dest, dtype=np.float64, order="F")
    _lib.check(lib.mlpp_add_a_xxT(dp(x), x.size, dp(out), out.shape[0], out.shape[1], C.c_double(a)))
    return out


def test_xAx_symmetric_errors():
    with pytest.raises(ValueError, match="not square"):
        _xAx(np.zeros((2, 3)), np.zeros(2))
    with pytest.raises(ValueError, match="wrong size"):
        _xAx(np.zeros((3, 3)), np.zeros(2))
    with pytest.raises(ValueError, match="Expected square matrix"):
        _add_a_xxT(np.zeros(3), np.zeros((3, 4)), 0.6)


@pytest.mark.parametrize("n", [4, 14, 15, 1024])
def test_linear_algebra_helpers(oracle, n):
    rng = np.random.default_rng(n)
    A0 = rng.uniform(-1, 1, (n, n))
    A = (A0 + A0.T) / 2
    x = rng.uniform(-1, 1, n)
    expected = x @ A @ x
    assert abs(_xAx(A, x) - expected) <= abs(expected) * 1e-14
    assert abs(_xAx(A, x) - oracle.xAx_symmetric(A, x)) <= abs(expected) * 1e-14
    Au = np.triu(A) + np.tril(rng.uniform(-1, 1, (n, n)), -1)          # only the upper triangle is read
    assert _xAx(Au, x) == _xAx(A, x)
    assert np.linalg.norm(_xxT(x) - np.outer(x, x)) <= np.linalg.norm(np.outer(x, x)) * 1e-15
    expected = A0 + 0.6 * np.outer(x, x)
    assert np.linalg.norm(_add_a_xxT(x, A0, 0.6) - expected) <= np.linalg.norm(expected) * 1e-15


# ---- initialisers: same libstdc++ <random> calls as the reference => bit-identical to the oracle ----------------

@pytest.mark.parametrize("seed", [None, 42, 63413131])
@pytest.mark.parametrize("kind,cls", [("FORGY", clustering.Forgy), ("RANDOM_PARTITION", clustering.RandomPartition),
                                       ("KPP", clustering.KPP)])
def test_centroid_initialisers_match_oracle(oracle, kind, cls, seed):
    data, _ = oracle.testdata_two_gaussians(400)
    for K in (1, 2, 7):
        ours = cls()._run(data, K, seed)
        ref = oracle.init_centroids(getattr(oracle, kind), data, K, seed)
        assert np.array_equal(ours, ref)


def test_forgy_picks_distinct_samples_in_ascending_order(oracle):
    data, _ = oracle.testdata_mousie(200)
    c = clustering.Forgy()._run(data, 5, 1234)
    idx = [int(np.where((data == row).all(axis=1))[0][0]) for row in c]
    assert idx == sorted(idx) and len(set(idx)) == 5        # std::sample = selection sampling


def test_closest_centroid_matches_oracle(oracle):
    data, _ = oracle.testdata_two_gaussians(400)
    for kind, cls in (("FORGY", clustering.Forgy), ("KPP", clustering.KPP)):
        ours = clustering.ClosestCentroid(cls())._run(data, 3, 99)
        ref = oracle.init_closest_centroid(getattr(oracle, kind), data, 3, 99)
        assert np.array_equal(ours, ref)
        assert np.array_equal(ours.sum(axis=1), np.ones(400))


def test_fixed_centroids():
    c = np.arange(6.0).reshape(2, 3)
    data = np.zeros((10, 3))
    assert np.array_equal(clustering.FixedCentroids(c)._run(data, 2), c)
    with pytest.raises(ValueError, match="do not match"):
        clustering.FixedCentroids(c)._run(data, 3)


# ---- covariance processing and M-step closing arithmetic --------------------------------------------------------

@pytest.mark.parametrize("d", [1, 2, 5, 32])
def test_process_covariance(oracle, d):
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d))
    cov = A @ A.T / d + 0.5 * np.eye(d)
    inv, sd = _lib.process_covariance(cov)
    assert np.max(np.abs(inv - np.linalg.inv(cov))) <= 1e-12 * np.max(np.abs(inv))
    assert abs(sd - np.sqrt(np.linalg.det(cov))) <= 1e-12 * sd
    em = oracle.EM(1)
    em.set_parameters(np.zeros((1, d)), cov[None], np.ones(1))
    assert np.max(np.abs(inv - em.inverse_covariances[0])) <= 1e-13 * np.max(np.abs(inv))
    assert abs(sd - em.sqrt_dets[0]) <= 1e-14 * sd


def packed_statistics(X, R, shift):
    """numpy restatement of the documented statistics layout (include/mlhip.h): per component the packed lower
    triangle of sum_i r_ik xt_i xt_i^T, xt = [x - shift; 1]."""
    n, d = X.shape
    xt = np.hstack([X - shift, np.ones((n, 1))])
    ia, ib = np.tril_indices(d + 1)          # row-major lower triangle: (a, b), a >= b, at a(a+1)/2 + b
    phi = xt[:, ia] * xt[:, ib]
    return R.T @ phi                         # K x F


def finalize(d, K, stats, shift, n):
    pi, mu, S = np.empty(K), np.empty((K, d)), np.empty((K, d, d))
    stats = np.ascontiguousarray(stats)
    shift = np.ascontiguousarray(shift)
    _lib.check(lib.mlhip_em_finalize_statistics(d, K, dp(stats), dp(shift), C.c_double(n), dp(pi), dp(mu), dp(S)))
    return pi, mu, S


@pytest.mark.parametrize("d,K", [(2, 3), (5, 4), (32, 6)])
def test_finalize_statistics_matches_oracle_mstep(oracle, d, K):
    rng = np.random.default_rng(100 + d)
    n = 500
    X = 3 + rng.standard_normal((n, d)) * rng.uniform(0.5, 2, d)
    R = rng.dirichlet(np.ones(K), n)
    cnt = C.c_uint32()
    _lib.check(lib.mlhip_em_statistics_count(d, C.byref(cnt)))
    assert cnt.value == (d + 1) * (d + 2) // 2
    shift = X.mean(axis=0)
    pi, mu, S = finalize(d, K, packed_statistics(X, R, shift), shift, n)
    em = oracle.EM(K)
    em.set_responsibilities(R, d)
    em.maximisation_step(X)
    assert np.max(np.abs(pi - em.mixing_probabilities)) < 1e-14
    assert np.max(np.abs(mu - em.means)) < 1e-13 * np.max(np.abs(em.means))
    assert np.max(np.abs(S - em.covariances)) < 1e-12 * np.max(np.abs(em.covariances))
    assert np.array_equal(S, np.transpose(S, (0, 2, 1)))    # exactly symmetric
