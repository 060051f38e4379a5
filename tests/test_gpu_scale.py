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


@pytest.fixture(scope="module")
def headline():
    """The N=10M, d=32, K=64 block of BASELINE.json configs[2] / [3] (2.56 GB on the host), generated once for the module."""
    from ml_amd import synth
    mix = synth.Mixture(32, 64)
    X, comp = mix.sample(10_000_000)
    return mix, X, comp


def test_em_full_size_properties(ctx, headline):
    """N=10M, d=32, K=64 (BASELINE.json configs[2])."""
    from ml_amd import _lib
    n, d, K = 10_000_000, 32, 64
    mix, X, comp = headline
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


def test_config_D_as_eight_shards_matches_the_one_shard_run(ctx, headline):
    """BASELINE.json configs[3] -- N=10M, d=32, K=64 row-sharded over 8 GPUs, one statistics all-reduce per iteration -- at FULL
    size through the library's device group: 8 shards of 1.25M rows, here all on the one GPU of the box (the in-process
    fixed-order all-reduce stands in for RCCL; everything else -- sharding, per-shard kernels, closing arithmetic on every shard,
    the end-of-fit checksum exchange that holds the shards to bit-identical parameters -- is the 8-GPU code path). Against the
    one-shard run of the same block: the log-likelihood history of 6 iterations to 1e-12, a converge run with the same number
    of steps and bit-exact labels on all 10M rows."""
    from ml_amd import _lib
    n, d, K = 10_000_000, 32, 64
    mix, X, _ = headline
    grp = _lib.Context.group(8, device_ids=[0] * 8)
    try:
        g, s = _lib.Data(grp, X), _lib.Data(ctx, X)
        assert [g.shard_rows(i)[1] for i in range(8)] == [1_250_000] * 8
        _, cov = s.sample_covariance()
        _, cov_g = g.sample_covariance()
        assert np.max(np.abs(cov_g - cov)) <= 1e-12 * np.max(np.abs(cov))
        pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
        a = g.em_iterate(pi, mu, S, 6)
        b = s.em_iterate(pi, mu, S, 6)
        assert a[0] == b[0] == 6
        assert np.max(np.abs(a[6] - b[6]) / np.abs(b[6])) < 1e-12, (a[6], b[6])
        assert np.max(np.abs(a[4] - b[4])) <= 1e-10 * np.max(np.abs(b[4]))
        assert np.max(np.abs(a[5] - b[5])) <= 1e-9 * np.max(np.abs(b[5]))
        # converge run from the same start (tolerance 1e-8 on the mean log-likelihood): same steps, bit-exact labels everywhere
        a = g.em_iterate(pi, mu, S, 60, atol=1e-8)
        b = s.em_iterate(pi, mu, S, 60, atol=1e-8)
        assert a[1] and b[1] and a[0] == b[0], (a[0], b[0])
        assert abs(a[2] - b[2]) <= 1e-12 * abs(b[2])
        la, lb = g.em_labels(K), s.em_labels(K)
        assert np.array_equal(la, lb)
        g.close(); s.close()
    finally:
        grp.close()


def test_config_E_full_size_on_one_gpu_and_as_eight_shards(ctx, oracle):
    """BASELINE.json configs[4] -- K-means N=100M, d=8, K=256 on 8 GPUs -- at FULL size (6.4 GB block): 4 Lloyd steps on one
    context with the per-step invariants, the labels of a 1M-row prefix against the oracle's assignment_step, and the same
    steps as 8 shards of 12.5M rows through the device group (counts equal, inertia 1e-12, all 100M labels equal)."""
    from ml_amd import _lib, synth
    n, d, K = 100_000_000, 8, 256
    mix = synth.Mixture(d, K, seed=77, diagonal=True)
    X, _ = mix.sample(n)
    C0 = mix.means + 0.3 * np.random.default_rng(1).standard_normal((K, d))
    dt = _lib.Data(ctx, X)
    C, inertias, per_step = C0, [], []
    for step in range(4):
        inertia, changed, counts, C_new = dt.kmeans_step(C)
        assert counts.sum() == n and (changed == n if step == 0 else changed < n)
        inertias.append(inertia)
        per_step.append((inertia, changed, counts))
        C_prev, C = C, C_new
    assert all(b <= a * (1 + 1e-12) for a, b in zip(inertias, inertias[1:])), inertias   # Lloyd never increases inertia
    labels = dt.kmeans_labels()
    dist2 = dt.kmeans_distances()
    # the last assignment (against C_prev) on the first 1M rows: labels AND distances as the reference's assignment_step gives them
    m = 1_000_000
    km = oracle.KMeans(K)
    km.set_centroids(C_prev, m)
    km.assignment_step(np.ascontiguousarray(X[:m]))
    assert np.array_equal(labels[:m], km.labels)
    want = ((X[:m] - C_prev[km.labels]) ** 2).sum(axis=1)
    assert np.max(np.abs(dist2[:m] - want)) <= 1e-12 * np.max(want)
    for k in (0, 255):                                   # the returned means are the means of the assigned rows
        sel = X[labels == k]
        assert sel.shape[0] == per_step[-1][2][k]
        assert np.max(np.abs(sel.mean(axis=0) - C[k])) <= 1e-11 * max(1.0, np.max(np.abs(C[k])))
    dt.close()
    grp = _lib.Context.group(8, device_ids=[0] * 8)
    try:
        g = _lib.Data(grp, X)
        assert [g.shard_rows(i)[1] for i in range(8)] == [12_500_000] * 8
        Cg = C0
        for step in range(4):
            inertia, changed, counts, Cg = g.kmeans_step(Cg)
            ref = per_step[step]
            assert changed == ref[1] and np.array_equal(counts, ref[2])
            assert abs(inertia - ref[0]) <= 1e-12 * ref[0]
        assert np.max(np.abs(Cg - C)) <= 1e-13 * np.max(np.abs(C))
        assert np.array_equal(g.kmeans_labels(), labels)
        g.close()
    finally:
        grp.close()


@pytest.mark.parametrize("d,K", [(4, 3), (8, 20), (16, 9), (32, 16)])
def test_few_cluster_kmeans_takes_the_direct_form_kernel_with_the_same_bits(d, K, monkeypatch):
    """From 2^21 rows on, K-means with few clusters (K <= 16, K <= 24 at d <= 8) runs on the direct-form kernel with its copies of
    the LDS accumulator table instead of the matrix-core search (kmeans.hip launch_kmeans_assign): labels, distances, counts and new
    centroids are the same bits as the matrix-core kernel's (MLHIP_KMEANS=mfma), the labels of a prefix the oracle's."""
    from ml_amd import _lib
    from oracle import oracle_ctypes as oracle
    n = (1 << 21) + 777
    rng = np.random.default_rng(d * 100 + K)
    C = 3.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(C[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    C0 = C + 0.5 * rng.standard_normal((K, d))
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    a = dt.kmeans_step(C0)
    la, da = dt.kmeans_labels(), dt.kmeans_distances()
    it_a = dt.kmeans_iterate(C0, 4, 0.0)
    monkeypatch.setenv("MLHIP_KMEANS", "mfma")
    b = dt.kmeans_step(C0)
    assert np.array_equal(la, dt.kmeans_labels()) and np.array_equal(da, dt.kmeans_distances())
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and abs(a[0] - b[0]) <= 1e-13 * b[0]
    it_b = dt.kmeans_iterate(C0, 4, 0.0)
    assert it_a[0] == it_b[0] and np.array_equal(it_a[3], it_b[3]) and np.array_equal(it_a[4], it_b[4])
    m = 30000
    km = oracle.KMeans(K)
    km.set_centroids(C0, m)
    km.assignment_step(X[:m])
    assert np.array_equal(la[:m], km.labels)
    dt.close()
    ctx.close()
