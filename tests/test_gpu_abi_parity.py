"""Parity of the HIP path (through the C ABI, include/mlhip.h) against the golden fixtures and the CPU oracle.
Needs a GPU: run with `pytest -m gpu`.

Tolerances (DESIGN.md "Tolerances"): log-likelihood 1e-12 relative; mixing/means 1e-11, covariances 1e-10
relative (max-norm); responsibilities 1e-12 absolute; labels bit-exact."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

EM_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "em_onestep_*.npz"))
                  if "_diag_" not in os.path.basename(p))          # the diagonal extension: tests/test_gpu_diag.py
KM_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kmeans_onestep_*.npz")))


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


@pytest.mark.parametrize("case", EM_CASES)
def test_em_step_matches_golden(ctx, case):
    g = load_golden(case)
    X = g["X"]
    K = g["pi0"].size
    dt = _data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(g["pi0"], g["mu0"], g["Sigma0"])
    assert abs(ll - float(g["ll0"])) <= 1e-12 * abs(float(g["ll0"]))
    assert relerr(pi1, g["pi1"]) < 1e-11
    assert relerr(mu1, g["mu1"]) < 1e-11
    assert relerr(S1, g["Sigma1"]) < 1e-10
    R = dt.em_responsibilities(K)
    assert np.max(np.abs(R - g["R0"])) < 1e-12
    assert np.array_equal(dt.em_labels(K), g["labels0"])
    # E-step-only / M-step-only entry points give the same numbers: bit-identical where em_step runs the same two kernels,
    # to rounding where it runs the fused small-shape kernel (which normalises r = e_k / sum_k e_k, not exp(lw_k - lse)).
    ll2 = dt.em_expectation(g["pi0"], g["mu0"], g["Sigma0"])
    assert abs(ll2 - ll) <= 1e-15 * abs(ll)
    pi2, mu2, S2 = dt.em_maximisation(K)
    assert relerr(pi2, pi1) < 1e-14 and relerr(mu2, mu1) < 1e-14 and relerr(S2, S1) < 1e-13
    # (bit-identical only where em_step runs the very same two kernels: for d >= 12 with K <= 64 em_step lets the statistics
    # kernel normalise the log-responsibilities itself, r = e_k / sum_k e_k, while the split entry points go through lse)
    # M-step from caller-given responsibilities (maximise_first path) and from hard labels.
    pi3, mu3, S3 = dt.em_maximisation_from(g["R0"])
    assert relerr(pi3, g["pi1"]) < 1e-11 and relerr(mu3, g["mu1"]) < 1e-11 and relerr(S3, g["Sigma1"]) < 1e-10
    dt.close()


@pytest.mark.parametrize("case", EM_CASES)
def test_em_step_matches_oracle(ctx, oracle, case):
    g = load_golden(case)
    X = g["X"]
    K = g["pi0"].size
    em = oracle.EM(K)
    em.set_parameters(g["mu0"], g["Sigma0"], g["pi0"])
    em.expectation_step(X)
    ll_ref = em.log_likelihood
    R_ref = em.responsibilities
    em.calculate_labels()
    labels_ref = em.labels
    em.maximisation_step(X)
    dt = _data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(g["pi0"], g["mu0"], g["Sigma0"])
    assert abs(ll - ll_ref) <= 1e-12 * abs(ll_ref)
    assert relerr(pi1, em.mixing_probabilities) < 1e-11
    assert relerr(mu1, em.means) < 1e-11
    assert relerr(S1, em.covariances) < 1e-10
    assert np.max(np.abs(dt.em_responsibilities(K) - R_ref)) < 1e-12
    assert np.array_equal(dt.em_labels(K), labels_ref)
    dt.close()


def test_em_hard_labels_mstep(ctx, oracle):
    g = load_golden("em_onestep_d4_K3.npz")
    X, labels, K = g["X"], g["labels0"], 3
    R = np.zeros((X.shape[0], K))
    R[np.arange(X.shape[0]), labels] = 1
    em = oracle.EM(K)
    em.set_responsibilities(R, X.shape[1])
    em.maximisation_step(X)
    dt = _data(ctx, X)
    pi1, mu1, S1 = dt.em_maximisation_from_labels(labels, K)
    assert relerr(pi1, em.mixing_probabilities) < 1e-12
    assert relerr(mu1, em.means) < 1e-12
    assert relerr(S1, em.covariances) < 1e-10
    dt.close()


@pytest.mark.parametrize("case", EM_CASES)
def test_sample_covariance(ctx, oracle, case):
    X = load_golden(case)["X"]
    dt = _data(ctx, X)
    mean, cov = dt.sample_covariance()
    assert relerr(mean, X.mean(axis=0)) < 1e-13
    assert relerr(cov, oracle.sample_covariance(X)) < 1e-12
    assert relerr(cov, np.cov(X.T)) < 1e-12
    dt.close()


@pytest.mark.parametrize("case", KM_CASES)
def test_kmeans_step_matches_golden(ctx, oracle, case):
    g = load_golden(case)
    X, C0 = g["X"], g["C0"]
    n, K = X.shape[0], C0.shape[0]
    dt = _data(ctx, X)
    inertia, changed, counts, C1 = dt.kmeans_step(C0)
    assert changed == n                       # first call: everything counts as changed
    assert np.array_equal(dt.kmeans_labels(), g["labels0"])
    assert abs(inertia - float(g["inertia0"])) <= 1e-13 * float(g["inertia0"])
    assert np.array_equal(counts, g["counts0"].astype(float))
    assert relerr(C1, g["C1"]) < 1e-13
    # Same centroids again -> nothing changes; per-sample distances equal the oracle's point query bit for bit.
    inertia2, changed2 = dt.kmeans_assign(C0)
    assert changed2 == 0 and inertia2 == inertia
    km = oracle.KMeans(K)
    km.set_centroids(C0, n)
    d2 = dt.min_squared_distances(C0)
    ref = np.array([km.assign_label(X[i])[1] for i in range(0, n, 7)])
    assert np.max(np.abs(d2[::7] - ref)) <= 1e-15 * np.max(ref)
    dt.close()


def test_kmeans_update_is_bitwise_reproducible_and_exact(ctx):
    """The update sums are exact fixed-point integer sums: bit-identical run to run, and equal to the correctly rounded
    exact sum (math.fsum) up to the final conversion."""
    import math
    rng = np.random.default_rng(12)
    n, d, K = 50_000, 5, 7
    X = np.ascontiguousarray(rng.standard_normal((n, d)) * np.array([1e-3, 1.0, 50.0, 1e4, 3.0]) + np.array([0, 5, -7, 1e5, 0.1]))
    C0 = X[rng.choice(n, K, replace=False)]
    runs = []
    for _ in range(3):
        dt = _data(ctx, X)
        runs.append(dt.kmeans_step(C0))
        labels = dt.kmeans_labels()
        dt.close()
    for r in runs[1:]:
        assert r[0] == runs[0][0] and np.array_equal(r[2], runs[0][2]) and np.array_equal(r[3], runs[0][3])
    _, _, counts, C1 = runs[0]
    for k in range(K):
        sel = X[labels == k]
        assert counts[k] == sel.shape[0]
        for j in range(d):
            exact = math.fsum(sel[:, j]) / sel.shape[0]
            assert abs(C1[k, j] - exact) <= 4e-16 * abs(exact)


def test_kmeans_empty_cluster_goes_to_origin(ctx):
    rng = np.random.default_rng(5)
    X = rng.standard_normal((1000, 3))
    C0 = np.array([[0.0, 0, 0], [100.0, 100, 100]])     # second cluster attracts nothing
    dt = _data(ctx, X)
    inertia, changed, counts, C1 = dt.kmeans_step(C0)
    assert counts[1] == 0 and np.array_equal(C1[1], np.zeros(3))   # ML/KMeans.cpp:184
    assert counts[0] == 1000
    dt.close()


@pytest.mark.parametrize("n,d,K", [(1, 1, 1), (255, 5, 3), (257, 7, 17), (4096, 32, 64), (10007, 16, 16), (3000, 8, 256)])
def test_em_step_ragged_shapes(ctx, oracle, n, d, K):
    """Sizes that are not multiples of the tile / block sizes, padded dimensions, many components."""
    rng = np.random.default_rng(n + d + K)
    means = 3.0 * rng.standard_normal((K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)))
    mu0 = means + 0.2 * rng.standard_normal((K, d))
    S0 = np.stack([np.eye(d) * rng.uniform(0.8, 1.5) + 0.05 for _ in range(K)])
    pi0 = np.full(K, 1.0 / K)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    ll_ref, R_ref = em.log_likelihood, em.responsibilities
    dt = _data(ctx, X)
    ll = dt.em_expectation(pi0, mu0, S0)
    assert abs(ll - ll_ref) <= 1e-12 * abs(ll_ref)
    assert np.max(np.abs(dt.em_responsibilities(K) - R_ref)) < 1e-12
    if n >= 4 * K:   # otherwise some components are (nearly) empty and the M-step is degenerate
        em.maximisation_step(X)
        pi1, mu1, S1 = dt.em_maximisation(K)
        assert relerr(pi1, em.mixing_probabilities) < 1e-11
        assert relerr(mu1, em.means) < 1e-11
        assert relerr(S1, em.covariances) < 1e-10
    dt.close()


def test_strided_input_and_device_upload(ctx):
    from ml_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(3)
    big = rng.standard_normal((500, 10))
    X = np.ascontiguousarray(big[:, :6])
    dt1 = _data(ctx, X)
    # ld > d through the raw ABI
    h = C.c_void_p()
    _lib.check(_lib.lib.mlhip_data_upload(ctx.handle, _lib.dptr(big), 6, C.c_uint64(500), C.c_int64(10), C.byref(h)))
    s1 = dt1.shift
    out = np.empty(6)
    _lib.check(_lib.lib.mlhip_data_shift(h, _lib.dptr(out)))
    assert np.array_equal(s1, out)
    assert np.max(np.abs(s1 - X.mean(axis=0))) < 1e-15
    _lib.check(_lib.lib.mlhip_data_free(h))
    dt1.close()


def test_bad_arguments(ctx):
    from ml_amd import _lib
    X = np.zeros((10, 3))
    dt = _data(ctx, X)
    with pytest.raises(ValueError):
        dt.em_responsibilities(2)            # no E-step yet
    with pytest.raises(ValueError):
        dt.kmeans_labels()
    with pytest.raises(TypeError):
        _lib.Data(ctx, np.zeros((10, 3), dtype=np.float32))
    with pytest.raises(_lib.MlhipError):
        _lib.Data(ctx, np.zeros((10, 4097)))  # d > 4096 unsupported (129..4096: the plain kernels of generic_dim.hip)
    dt.close()


@pytest.mark.parametrize("d,K", [(12, 7), (16, 16), (20, 5), (24, 9), (28, 3), (32, 64)])
def test_estep_kernel_variants_agree(ctx, oracle, d, K, monkeypatch):
    """The two E-step kernels of the default build that exist for d in 12..32 (4x4x4 MFMA = default, scalar-fed VALU) compute
    the same log-likelihood / responsibilities / labels; the default one is also checked against the oracle. (The 16x16x4
    variant is an experiment: `make EXPERIMENTS=1`.)"""
    rng = np.random.default_rng(d * 100 + K)
    n = 3000
    means = 2.5 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    mu0 = means + 0.3 * rng.standard_normal((K, d))
    S0 = np.stack([np.cov(X.T) * rng.uniform(0.2, 0.6) + 0.1 * np.eye(d) for _ in range(K)])
    pi0 = rng.dirichlet(np.ones(K) * 5)
    out = {}
    for variant in ("", "valu"):
        if variant:
            monkeypatch.setenv("MLHIP_ESTEP", variant)
        else:
            monkeypatch.delenv("MLHIP_ESTEP", raising=False)
        dt = _data(ctx, X)
        ll = dt.em_expectation(pi0, mu0, S0)
        out[variant] = (ll, dt.em_responsibilities(K), dt.em_labels(K), dt.em_maximisation(K))
        dt.close()
    monkeypatch.delenv("MLHIP_ESTEP", raising=False)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    ll0, R0, lab0, m0 = out[""]
    assert abs(ll0 - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    assert np.max(np.abs(R0 - em.responsibilities)) < 1e-12
    for variant in ("valu",):
        ll, R, lab, m = out[variant]
        assert abs(ll - ll0) <= 1e-13 * abs(ll0)
        assert np.max(np.abs(R - R0)) < 1e-12
        assert np.array_equal(lab, lab0)
        for a, b in zip(m, m0):
            assert np.max(np.abs(a - b)) <= 1e-11 * np.max(np.abs(b))


@pytest.mark.parametrize("n,d,K", [(5000, 4, 3), (20000, 8, 256), (7001, 8, 17), (3000, 16, 40), (3000, 32, 64), (2000, 12, 1),
                                   (3000, 40, 9), (2500, 64, 33), (3000, 50, 160), (6000, 16, 256), (9000, 8, 1500),
                                   (5000, 32, 700), (4000, 64, 300), (70000, 4, 5000), (3000, 100, 70), (2000, 128, 260), (30000, 8, 10000), (20000, 2, 4000), (9000, 3, 256), (9000, 6, 300), (5000, 5, 130), (4000, 1, 128)])
def test_kmeans_mfma_and_valu_kernels_agree_bitwise(ctx, oracle, n, d, K, monkeypatch):
    """The matrix-core search (approximate scores + exact recheck / exact fallback) yields the same labels and the same
    per-sample distances, bit for bit, as the direct-form VALU kernel and as the oracle's point query -- including on
    data with exact ties (duplicated centroids, samples equidistant from two centroids)."""
    rng = np.random.default_rng(n + d + K)
    C = 2.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(C[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    if K >= 3:
        C[2] = C[0]                                   # duplicate centroid: exact tie for every sample near it
        X[:50] = 0.5 * (C[0] + C[1])                  # exactly equidistant samples (up to rounding of the midpoint)
        X[50:60] = C[1]                               # zero distance
    res = {}
    for variant in ("mfma", "valu"):
        if variant == "valu":
            monkeypatch.setenv("MLHIP_KMEANS", "valu")
        else:
            monkeypatch.delenv("MLHIP_KMEANS", raising=False)
        dt = _data(ctx, X)
        inertia, changed, counts, C1 = dt.kmeans_step(C)
        res[variant] = (inertia, counts, C1, dt.kmeans_labels(), dt.min_squared_distances(C))
        dt.close()
    monkeypatch.delenv("MLHIP_KMEANS", raising=False)
    a, b = res["mfma"], res["valu"]
    assert np.array_equal(a[3], b[3])                 # labels
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    km = oracle.KMeans(K)
    km.set_centroids(C, n)
    km.assignment_step(X)
    assert np.array_equal(a[3], km.labels)
    step = max(1, n // 500)
    ref = np.array([km.assign_label(X[i])[1] for i in range(0, n, step)])
    assert np.array_equal(a[4][::step], ref)          # distances bit-identical to the oracle's fma chain


@pytest.mark.parametrize("n,d,K", [(6000, 8, 256), (6000, 4, 70), (5000, 8, 2100), (5000, 16, 300), (4000, 32, 520),
                                   (6000, 4, 5000), (4000, 12, 200)])
def test_kmeans_quad_tracking_sees_the_winners_mates(ctx, oracle, n, d, K, monkeypatch):
    """For K >= 16 d the scoring loop tracks the maximum of each accumulator's four scores (clusters k, k+4, k+8, k+12 of a
    16-block) and the exact phase scores the winner's three quad-mates itself. Ties and near-ties INSIDE a quad --
    duplicated mates, samples on and next to the bisector of two mates, in the first block, in the padded last block
    and beyond the first 1024-cluster sub-chunk -- must still give the reference's label and distance bit for bit."""
    rng = np.random.default_rng(7 * n + d + K)
    C = 2.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(C[rng.integers(0, K, n)] + 0.7 * rng.standard_normal((n, d)))
    last = 16 * ((K - 1) // 16)                       # first cluster of the last (possibly padded) block
    pairs = [(0, 4), (1, 13), (18, 26), (last, last + 4 if last + 4 < K else last + 1), (K - 1, K - 5)]
    if K > 1100:
        pairs += [(1030, 1038), (1024 + 16 * 3 + 2, 1024 + 16 * 3 + 14)]
    C[pairs[0][1]] = C[pairs[0][0]]                   # a duplicated quad-mate: exact tie for every sample near it
    X[:40] = C[pairs[0][0]] + 0.01 * rng.standard_normal((40, d))
    row = 40
    for a, b in pairs[1:]:
        mid = 0.5 * (C[a] + C[b])
        X[row:row + 8] = mid                          # on the bisector (up to the rounding of the midpoint)
        for e in (1e-15, 1e-13, 1e-11, 1e-8):         # and next to it, on either side
            X[row + 8] = mid + e * (C[a] - C[b])
            X[row + 9] = mid - e * (C[a] - C[b])
            row += 2
        row += 8
    res = {}
    ctx.timing_enable(False)
    for variant in ("mfma", "valu"):
        if variant == "valu":
            monkeypatch.setenv("MLHIP_KMEANS", "valu")
        else:
            monkeypatch.delenv("MLHIP_KMEANS", raising=False)
        dt = _data(ctx, X)
        inertia, changed, counts, C1 = dt.kmeans_step(C)
        res[variant] = (inertia, counts, C1, dt.kmeans_labels(), dt.min_squared_distances(C))
        dt.close()
    monkeypatch.delenv("MLHIP_KMEANS", raising=False)
    a, b = res["mfma"], res["valu"]
    km = oracle.KMeans(K)
    km.set_centroids(C, n)
    km.assignment_step(X)
    assert np.array_equal(a[3], km.labels)
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    ref = np.array([km.assign_label(X[i])[1] for i in range(0, row)])
    assert np.array_equal(a[4][:row], ref)            # the contested samples' distances, bit-identical to the oracle


@pytest.mark.parametrize("d", [8, 4])        # d = 4: the fused small-shape kernel (the refinement rebuilds lw first)
@pytest.mark.parametrize("sigma,tol", [(1.0, 1e-12), (1e-2, 1e-11), (1e-4, 1e-10), (1e-6, 1e-8)])
def test_tight_clusters_far_from_the_global_mean(ctx, oracle, sigma, tol, d):
    """Clusters whose width is tiny next to their distance from the global mean: the one-GEMM statistics (shared shift)
    would cancel log10((spread/sigma)^2) digits of the covariances, so such components get a second pass about their own
    mean. Parity with the reference's two-pass form holds down to the conditioning of the problem itself
    (~1e-16 * spread / sigma); well-conditioned components are never refined."""
    rng = np.random.default_rng(5)
    K, n = 4, 4000
    means = 10.0 * rng.standard_normal((K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + sigma * rng.standard_normal((n, d)))
    mu0 = means + 0.1 * sigma * rng.standard_normal((K, d))
    S0 = np.stack([np.eye(d) * sigma ** 2] * K)
    pi0 = np.full(K, 1.0 / K)
    ctx.timing_enable(True)
    ctx.timing_reset()
    dt = _data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(pi0, mu0, S0)
    dt.close()
    _, refinements = ctx.timing_get("em_refine")
    ctx.timing_enable(False)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    em.maximisation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    assert relerr(mu1, em.means) < 1e-13
    assert relerr(S1, em.covariances) < tol
    assert refinements == (0 if sigma == 1.0 else K)


@pytest.mark.parametrize("d,K,n", [(1, 2, 900), (2, 8, 5000), (3, 16, 4000), (4, 64, 9000), (5, 20, 3000), (6, 32, 7001)])
def test_fused_small_shape_step_agrees_with_two_kernel_path(ctx, oracle, d, K, n, monkeypatch):
    """Small shapes run E-step + statistics as ONE kernel (no N x K block in HBM). Same numbers as the two-kernel path to
    rounding, same labels / responsibilities afterwards (rebuilt on demand from the same parameter records), and the
    oracle's values within the usual tolerances."""
    from ml_amd import synth
    mix = synth.Mixture(d, K, seed=100 + d)
    X, _ = mix.sample(n)
    pi0, mu0 = np.full(K, 1.0 / K), mix.initial_means()
    S0 = np.stack([np.cov(X.T).reshape(d, d)] * K)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("MLHIP_FUSED", fused)
        ctx.timing_enable(True)
        ctx.timing_reset()
        dt = _data(ctx, X)
        step = dt.em_step(pi0, mu0, S0)
        _, n_fused = ctx.timing_get("em_fused")
        res[fused] = (step, dt.em_labels(K), dt.em_responsibilities(K))
        ctx.timing_enable(False)
        dt.close()
        assert (n_fused == 1) == (fused == "1")
    monkeypatch.delenv("MLHIP_FUSED", raising=False)
    (a, la, ra), (b, lb, rb) = res["1"], res["0"]
    assert abs(a[0] - b[0]) <= 1e-14 * abs(b[0])
    assert relerr(a[1], b[1]) < 1e-13 and relerr(a[2], b[2]) < 1e-13 and relerr(a[3], b[3]) < 1e-12
    assert np.array_equal(la, lb) and np.array_equal(ra, rb)        # both rebuilt / built by the same E-step kernel
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(a[0] - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    assert np.max(np.abs(ra - em.responsibilities)) < 1e-12
    em.maximisation_step(X)
    assert relerr(a[1], em.mixing_probabilities) < 1e-11 and relerr(a[2], em.means) < 1e-11
    assert relerr(a[3], em.covariances) < 1e-9


def test_distance_probe_leaves_the_last_assignment_alone(ctx):
    """mlhip_min_squared_distances (the K-means++ weights pass) must not overwrite what mlhip_kmeans_distances / labels report
    for the last assignment."""
    rng = np.random.default_rng(4)
    X = rng.standard_normal((5000, 8))
    C = rng.standard_normal((6, 8))
    dt = _data(ctx, X)
    dt.kmeans_step(C)
    before, labels = dt.kmeans_distances(), dt.kmeans_labels()
    probe = dt.min_squared_distances(C[:1] + 3.0)
    assert np.array_equal(probe, ((X - (C[:1] + 3.0)) ** 2).sum(axis=1)) or np.allclose(probe, ((X - (C[:1] + 3.0)) ** 2).sum(axis=1), rtol=1e-14)
    assert np.array_equal(dt.kmeans_distances(), before)
    assert np.array_equal(dt.kmeans_labels(), labels)
    _, changed = dt.kmeans_assign(C)
    assert changed == 0                                     # the label history is intact too
    dt.close()


def test_kmeans_refuses_non_finite_data(ctx):
    X = np.random.default_rng(5).standard_normal((1000, 4))
    X[17, 2] = np.inf
    dt = _data(ctx, X)
    with pytest.raises(ValueError):
        dt.kmeans_step(np.zeros((3, 4)))
    dt.close()
