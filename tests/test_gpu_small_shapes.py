"""The vector-unit form of the fused E+M kernel (em_fused_small.hip: em_fused_valu_kernel) -- what EM::expectation_step and
EM::maximisation_step (reference ML/EM.cpp:196-247) become at d <= 6 with few components, the regime of the reference's own
benchmark (Benchmarks/bm_EM.cpp: d = 2, K = 3) -- against the oracle, against the matrix-core form of the same kernel
(MLHIP_FUSED_VALU=0) and against itself (run-to-run bit-reproducible). Every (d, K) border is covered: K = 1, the largest K
taken at every sample count (K F <= 64), the largest K the form is built for (taken from 2^20 samples on), one above each (the
last must fall to the matrix-core form and still agree), and d = 5 (padded to 6: matrix-core form)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAXK = {1: 32, 2: 16, 3: 10, 4: 7, 6: 4}         # valu_max_k (em_fused_small.hip)
ALWAYS = {1: 21, 2: 10, 3: 6, 4: 4, 6: 2}        # largest K with K F <= 64: the form is taken at every sample count


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


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
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)) + 40.0)    # off-centre: the shift matters
    mu0 = means + 40.0 + 0.3 * rng.standard_normal((K, d))
    S0 = np.stack([np.eye(d) * rng.uniform(0.7, 1.6) + 0.1 for _ in range(K)])
    pi0 = rng.uniform(0.5, 1.5, K)
    return X, pi0 / pi0.sum(), mu0, S0


SHAPES = [(d, K) for d in MAXK for K in sorted({1, 2, 3, ALWAYS[d], ALWAYS[d] + 1, MAXK[d], MAXK[d] + 1})] + [(5, 3)]


@pytest.mark.parametrize("d,K", SHAPES)
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 300007, (1 << 20) + 77])
def test_step_against_the_oracle_and_the_matrix_core_form(ctx, oracle, d, K, n, monkeypatch):
    from ml_amd import _lib
    X, pi0, mu0, S0 = _problem(d, K, n, 1000 * d + 10 * K + n % 97)
    dt = _lib.Data(ctx, X)
    monkeypatch.setenv("MLHIP_FUSED_VALU", "1")
    got = dt.em_step(pi0, mu0, S0)
    R = dt.em_responsibilities(K)
    labels = dt.em_labels(K)
    again = dt.em_step(pi0, mu0, S0)
    assert got[0] == again[0] and all(np.array_equal(a, b) for a, b in zip(got[1:], again[1:]))     # reproducible
    monkeypatch.setenv("MLHIP_FUSED_VALU", "0")
    mc = dt.em_step(pi0, mu0, S0)
    R_mc = dt.em_responsibilities(K)
    # the two forms differ only in the order of the sums over samples
    assert abs(got[0] - mc[0]) <= 1e-13 * abs(mc[0])
    assert np.max(np.abs(R - R_mc)) < 1e-14
    # oracle: E-step always; M-step when no component is (nearly) empty
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(got[0] - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    assert np.max(np.abs(R - em.responsibilities)) < 1e-12
    sure = np.sort(em.responsibilities, axis=1)
    sure = sure[:, -1] - (sure[:, -2] if K > 1 else 0.0) > 1e-9
    assert np.array_equal(labels[sure], np.argmax(em.responsibilities, axis=1)[sure])
    if n >= 50 * K:
        em.maximisation_step(X)
        for a, b, c, tol in zip(got[1:], mc[1:], (em.mixing_probabilities, em.means, em.covariances), (1e-12, 1e-12, 1e-10)):
            assert relerr(a, c) < tol
            assert relerr(a, b) < tol
    dt.close()


@pytest.mark.parametrize("d,K,n", [(2, 3, 200000), (2, 8, 50000), (1, 16, 30000), (3, 6, 40000), (4, 4, 40000), (6, 2, 30000),
                                   (6, 4, 1100000), (4, 7, 1050000), (2, 16, 1048576)])
def test_fit_loop_on_the_vector_unit_form(ctx, oracle, d, K, n, monkeypatch):
    """The whole loop (mlhip_em_iterate: lagged convergence test, device closing) over the vector-unit kernel: same number of
    steps and the same fit as over the matrix-core form, and as the oracle's EM::fit loop from the same start."""
    from ml_amd import _lib
    X, pi0, mu0, S0 = _problem(d, K, n, 7 * d + K)
    dt = _lib.Data(ctx, X)
    monkeypatch.setenv("MLHIP_FUSED_VALU", "1")
    got = dt.em_iterate(pi0, mu0, S0, 40, atol=1e-8)
    monkeypatch.setenv("MLHIP_FUSED_VALU", "0")
    mc = dt.em_iterate(pi0, mu0, S0, 40, atol=1e-8)
    assert got[0] == mc[0] and got[1] == mc[1]
    assert abs(got[2] - mc[2]) <= 1e-11 * abs(mc[2])
    for a, b in zip(got[3:6], mc[3:6]):
        assert relerr(a, b) < 1e-9
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    old = None
    for step in range(40):
        em.expectation_step(X)
        ll = em.log_likelihood
        em.maximisation_step(X)
        if old is not None and abs(ll - old) < 1e-8:
            break
        old = ll
    assert got[0] == step + 1
    assert abs(got[2] - ll) <= 1e-11 * abs(ll)
    assert relerr(got[4], em.means) < 1e-9
    dt.close()


@pytest.mark.parametrize("d,K", [(7, 1), (7, 5), (8, 3), (8, 16), (8, 17), (7, 32), (8, 32), (6, 20), (5, 32)])
@pytest.mark.parametrize("n", [1, 65, 4097, 200003])
def test_fused_step_with_scalar_fed_records(ctx, oracle, d, K, n, monkeypatch):
    """d = 7, 8 (K <= 32): the fused E+M kernel with the component records in scalar registers (em_fused_small.hip, SFEED) against
    the two-kernel path (MLHIP_FUSED=0: E-step + statistics kernels) and the oracle; d = 5, 6 take the same feed with
    MLHIP_FUSED_SFEED=1 (an A/B switch) and must agree with their LDS-fed default bit for bit (same arithmetic, same order)."""
    from ml_amd import _lib
    X, pi0, mu0, S0 = _problem(d, K, n, 31 * d + K + n % 89)
    dt = _lib.Data(ctx, X)
    if d <= 6:
        ref = dt.em_step(pi0, mu0, S0)
        monkeypatch.setenv("MLHIP_FUSED_SFEED", "1")
    got = dt.em_step(pi0, mu0, S0)
    R = dt.em_responsibilities(K)
    if d <= 6:
        assert got[0] == ref[0] and all(np.array_equal(a, b) for a, b in zip(got[1:], ref[1:]))
    monkeypatch.setenv("MLHIP_FUSED", "0")
    two = dt.em_step(pi0, mu0, S0)
    assert abs(got[0] - two[0]) <= 1e-13 * abs(two[0])
    assert np.max(np.abs(R - dt.em_responsibilities(K))) < 1e-13
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(got[0] - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    assert np.max(np.abs(R - em.responsibilities)) < 1e-12
    if n >= 50 * K:
        em.maximisation_step(X)
        for a, b, c, tol in zip(got[1:], two[1:], (em.mixing_probabilities, em.means, em.covariances), (1e-12, 1e-12, 1e-10)):
            assert relerr(a, c) < tol
            assert relerr(a, b) < tol
    dt.close()
