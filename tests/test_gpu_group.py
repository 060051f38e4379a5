"""Device group (mlhip_ctx_create_group): ONE context, the caller's one d x N block row-sharded over n shards inside the library --
the single-process multi-GPU form of `bool EM::fit(Eigen::Ref<const MatrixXd>)` (reference ML/EM.cpp:91) and KMeans::fit
(ML/KMeans.cpp:25). On a one-GPU box all shards sit on GPU 0 (the in-process all-reduce sums them in shard order); every call
through the group must return what a single context returns on the same block -- up to the summation order of the statistics
(log-likelihood 1e-12, parameters 1e-10, labels bit-exact) -- and the shards must hold bit-identical parameters (the library's
own end-of-fit checksum exchange fails the call otherwise)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def single():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def group3():
    from ml_amd import _lib
    c = _lib.Context.group(3, device_ids=[0, 0, 0])
    yield c
    c.close()


def _mixture(d, K, n, seed, spread=3.0):
    rng = np.random.default_rng(seed)
    means = spread * rng.standard_normal((K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)))
    mu0 = means + 0.2 * rng.standard_normal((K, d))
    S0 = np.broadcast_to(np.cov(X.T).reshape(d, d), (K, d, d)).copy()
    return X, np.full(K, 1.0 / K), mu0, S0


def test_group_reports_its_shape(group3, single):
    from ml_amd import _lib
    assert group3.shards == 3 and group3.shard_devices == [0, 0, 0]
    assert group3.reduce_kind == "group-direct"            # several shards on one GPU: the in-process sum
    assert group3.world == (1, 0)                          # one rank to its caller
    assert single.shards == 1 and single.reduce_kind == "none"
    X = np.random.default_rng(0).standard_normal((1000, 5))
    dt = _lib.Data(group3, X)
    assert dt.n_global == 1000
    rows = [dt.shard_rows(s) for s in range(3)]
    assert rows == [(0, 334), (334, 333), (667, 333)]      # the split of ml_amd.dist.shard_bounds
    ref = _lib.Data(single, X)
    assert np.array_equal(dt.shift, ref.shift) or relerr(dt.shift, ref.shift) < 1e-13
    with pytest.raises(_lib.MlhipError):
        group3.set_allreduce(lambda *a: None, False, 2, 0)  # a group takes no hook of its own
    dt.close(); ref.close()


@pytest.mark.parametrize("d,K,n", [(16, 8, 20003), (32, 16, 9001), (4, 3, 12001), (8, 5, 7000), (2, 3, 5)])
def test_group_em_step_and_outputs_match_a_single_context(group3, single, d, K, n):
    from ml_amd import _lib
    X, pi, mu, S = _mixture(d, K, n, 100 + d)
    g, s = _lib.Data(group3, X), _lib.Data(single, X)
    ll_g, pi_g, mu_g, S_g = g.em_step(pi, mu, S)
    ll_s, pi_s, mu_s, S_s = s.em_step(pi, mu, S)
    assert abs(ll_g - ll_s) <= 1e-12 * abs(ll_s)
    assert relerr(pi_g, pi_s) < 1e-11 and relerr(mu_g, mu_s) < 1e-11 and relerr(S_g, S_s) < 1e-10
    # the whole-sample arrays come back in the caller's row order, shard by shard
    assert np.array_equal(g.em_labels(K), s.em_labels(K))
    R_g, R_s = g.em_responsibilities(K), s.em_responsibilities(K)
    assert np.max(np.abs(R_g - R_s)) < 1e-12
    lo, cnt = max(0, n // 3 - 4), min(n, 9)               # a row range across a shard boundary
    assert np.array_equal(g.em_responsibilities_rows(K, lo, cnt), R_g[lo:lo + cnt])
    assert np.array_equal(s.em_responsibilities_rows(K, lo, cnt), R_s[lo:lo + cnt])
    # E-step / M-step as separate calls, and the maximise-first entries on whole-sample inputs
    assert abs(g.em_expectation(pi, mu, S) - s.em_expectation(pi, mu, S)) <= 1e-12 * abs(ll_s)
    for a, b in zip(g.em_maximisation(K), s.em_maximisation(K)):
        assert relerr(a, b) < 1e-10
    labels = np.random.default_rng(5).integers(0, K, n).astype(np.uint32)
    labels[:K] = np.arange(K)
    for a, b in zip(g.em_maximisation_from_labels(labels, K), s.em_maximisation_from_labels(labels, K)):
        assert relerr(a, b) < 1e-10
    R = np.asfortranarray(np.random.default_rng(6).dirichlet(np.ones(K), n))
    for a, b in zip(g.em_maximisation_from(R), s.em_maximisation_from(R)):
        assert relerr(a, b) < 1e-10
    m_g, c_g = g.sample_covariance()
    m_s, c_s = s.sample_covariance()
    assert relerr(m_g, m_s) < 1e-12 and relerr(c_g, c_s) < 1e-11
    g.close(); s.close()


@pytest.mark.parametrize("d,K,n,diagonal", [
    (32, 16, 30001, False),     # matrix-core E-step + wide statistics kernel, synchronous or lagged by the GLOBAL size
    (16, 5, 9000, False),
    (4, 3, 10000, False),       # fused small-shape kernel, lagged loop: speculative all-reduces of all shards stay paired
    (16, 16, 30000, True),      # diagonal covariances
    (100, 3, 4001, False),      # host closing (d > 64)
    (200, 3, 4001, False),      # d > 128: the matrix-core tier of big_dim.hip in every shard
    (8, 20, 30000, False),      # fused kernel, records from scalar registers
    (2, 3, 50000, False),       # vector-unit statistics
])
def test_group_em_iterate_matches_a_single_context(group3, single, d, K, n, diagonal):
    from ml_amd import _lib
    X, pi, mu, S = _mixture(d, K, n, 7 + d)
    if diagonal:
        S = np.stack([np.diag(m) for m in S])
    g, s = _lib.Data(group3, X), _lib.Data(single, X)
    out_g = g.em_iterate(pi, mu, S, 40, atol=1e-9, rtol=0.0, diagonal=diagonal)
    out_s = s.em_iterate(pi, mu, S, 40, atol=1e-9, rtol=0.0, diagonal=diagonal)
    assert out_g[0] == out_s[0] and out_g[1] == out_s[1]                    # same steps, same verdict
    assert np.max(np.abs(out_g[6] - out_s[6]) / np.abs(out_s[6])) < 1e-12    # the log-likelihood trajectory
    assert relerr(out_g[3], out_s[3]) < 1e-10 and relerr(out_g[4], out_s[4]) < 1e-10 and relerr(out_g[5], out_s[5]) < 1e-9
    assert np.array_equal(g.em_labels(K), s.em_labels(K))
    g.close(); s.close()


@pytest.mark.parametrize("d,K,n", [(8, 32, 50001), (2, 4, 20000), (32, 7, 9000), (3, 130, 20000), (200, 24, 6000)])
def test_group_kmeans_matches_a_single_context(group3, single, d, K, n):
    from ml_amd import _lib
    rng = np.random.default_rng(11 + d)
    X = np.ascontiguousarray(rng.standard_normal((n, d)) + 4.0 * rng.integers(0, 3, (n, 1)))
    C0 = X[rng.choice(n, K, replace=False)].copy()
    g, s = _lib.Data(group3, X), _lib.Data(single, X)
    i_g, ch_g, cnt_g, c_g = g.kmeans_step(C0)
    i_s, ch_s, cnt_s, c_s = s.kmeans_step(C0)
    assert ch_g == ch_s == n and np.array_equal(cnt_g, cnt_s)
    assert abs(i_g - i_s) <= 1e-12 * i_s and relerr(c_g, c_s) < 1e-13
    assert np.array_equal(g.kmeans_labels(), s.kmeans_labels())
    assert np.array_equal(g.kmeans_distances(), s.kmeans_distances())        # per-sample distances are bit-identical
    out_g, out_s = g.kmeans_iterate(C0, 25, atol=1e-9), s.kmeans_iterate(C0, 25, atol=1e-9)
    assert out_g[0] == out_s[0] and out_g[1] == out_s[1]
    assert abs(out_g[2] - out_s[2]) <= 1e-12 * out_s[2] and np.array_equal(out_g[3], out_s[3])
    assert relerr(out_g[4], out_s[4]) < 1e-12
    assert np.array_equal(g.kmeans_labels(), s.kmeans_labels())
    assert np.array_equal(g.min_squared_distances(C0[:3]), s.min_squared_distances(C0[:3]))
    ia_g, ia_s = g.kmeans_assign(out_s[4]), s.kmeans_assign(out_s[4])
    assert abs(ia_g[0] - ia_s[0]) <= 1e-12 * ia_s[0] and ia_g[1] == ia_s[1]
    g.close(); s.close()


def test_group_of_eight_shards_with_tiny_and_empty_shards(single):
    """More shards than rows in some of them: a shard may hold one row or none and still joins every collective."""
    from ml_amd import _lib
    ctx = _lib.Context.group(8, device_ids=[0] * 8)
    try:
        X, pi, mu, S = _mixture(3, 2, 5, 3)                # 5 rows over 8 shards: three are empty
        X = np.ascontiguousarray(np.vstack([X, X + 0.5, X - 0.25]))[:13]
        g, s = _lib.Data(ctx, X), _lib.Data(single, X)
        assert [g.shard_rows(i)[1] for i in range(8)] == [2, 2, 2, 2, 2, 1, 1, 1]
        a, b = g.em_step(pi, mu, S), s.em_step(pi, mu, S)
        assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and relerr(a[3], b[3]) < 1e-10
        assert np.array_equal(g.em_labels(2), s.em_labels(2))
        g.close()
        Y = X[:5]
        g = _lib.Data(ctx, Y)
        r = _lib.Data(single, Y)
        assert [g.shard_rows(i)[1] for i in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
        a, b = g.em_step(pi, mu, S), r.em_step(pi, mu, S)
        assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
        ka, kb = g.kmeans_step(Y[:2].copy()), r.kmeans_step(Y[:2].copy())
        assert np.array_equal(ka[2], kb[2]) and relerr(ka[3], kb[3]) < 1e-13
        assert np.array_equal(g.kmeans_labels(), r.kmeans_labels())
        g.close(); r.close(); s.close()
    finally:
        ctx.close()


def test_a_failing_call_leaves_the_group_usable(group3):
    """An argument error is the same on every shard; a later call on the same group works (the shards resynchronise)."""
    from ml_amd import _lib
    X, pi, mu, S = _mixture(4, 3, 3000, 1)
    g = _lib.Data(group3, X)
    with pytest.raises(ValueError):
        g.em_iterate(pi, mu, S, 5, atol=-1.0)
    with pytest.raises((ValueError, _lib.MlhipError)):
        g.em_labels(3)                                      # no E-step yet
    out = g.em_iterate(pi, mu, S, 5)
    assert out[0] == 5 and np.all(np.isfinite(out[6]))
    g.close()


def test_group_fits_carry_the_allreduce_timer_and_the_shard_checksum_guard(group3, monkeypatch):
    """What a first run on real multi-GPU hardware needs from the group path: the all-reduce is timed on every shard's stream,
    the route is named, and shards whose sums differ fail the fit (the ranks' end-of-fit checksum exchange, through the group)."""
    from ml_amd import _lib
    X, pi, mu, S = _mixture(12, 6, 20000, 9)
    g = _lib.Data(group3, X)
    group3.timing_reset()
    group3.timing_enable(True)
    out = g.em_iterate(pi, mu, S, 6)
    ms, launches = group3.timing_get("allreduce")
    group3.timing_enable(False)
    assert out[0] == 6 and launches >= 6 and ms > 0.0
    assert group3.reduce_kind == "group-direct" and group3.rccl_ranks == 0      # (group-rccl reports ncclCommCount here)
    assert group3.timing_get("em_estep")[1] >= 6
    g.close()
    # a group whose shard 1 returns sums that differ by 1e-12 relative in one entry (the hook is read when the group is created)
    monkeypatch.setenv("MLHIP_GROUP_TEST_PERTURB", "1")
    bad = _lib.Context.group(3, device_ids=[0, 0, 0])
    monkeypatch.delenv("MLHIP_GROUP_TEST_PERTURB")
    try:
        gb = _lib.Data(bad, X)
        with pytest.raises(_lib.MlhipError, match="ranks disagree"):
            gb.em_iterate(pi, mu, S, 4)
        C0 = X[:5].copy()
        with pytest.raises(_lib.MlhipError, match="ranks disagree"):
            gb.kmeans_iterate(C0, 4)
        gb.close()
    finally:
        bad.close()


@pytest.mark.parametrize("fail_at", [1, 2, 7])
def test_one_shard_failing_alone_fails_the_call_and_the_group_recovers(single, monkeypatch, fail_at):
    """a shard that fails BY ITSELF (out of memory, a HIP error -- here: injected into its fail_at-th all-reduce; the
    first one is also the slots' growth) while the others are inside the same all-reduce: the call comes back with that shard's
    error on the caller's thread instead of hanging, and the next call -- after every shard has drained its stream and dropped its
    slots -- gives the results of an undisturbed group."""
    from ml_amd import _lib
    X, pi, mu, S = _mixture(6, 4, 12000, 5)
    monkeypatch.setenv("MLHIP_GROUP_TEST_FAIL", f"2:{fail_at}")
    grp = _lib.Context.group(4, device_ids=[0, 0, 0, 0])
    monkeypatch.delenv("MLHIP_GROUP_TEST_FAIL")
    try:
        with pytest.raises(_lib.MlhipError, match="injected failure"):
            g = _lib.Data(grp, X)                            # (the upload has all-reduces of its own: fail_at = 1 hits the first of them)
            for _ in range(4):
                g.em_iterate(pi, mu, S, 5)
        g = _lib.Data(grp, X)
        s = _lib.Data(single, X)
        a, b = g.em_iterate(pi, mu, S, 6), s.em_iterate(pi, mu, S, 6)
        assert a[0] == b[0] and relerr(a[6], b[6]) < 1e-12 and relerr(a[4], b[4]) < 1e-11
        assert np.array_equal(g.em_labels(4), s.em_labels(4))
        g.close(); s.close()
    finally:
        grp.close()


def test_group_on_distinct_gpus_when_there_are_two():
    """Shards on distinct GPUs: RCCL communicators from ncclCommInitAll (or, MLHIP_GROUP_REDUCE=direct, peer access). Skipped on a
    one-GPU box."""
    from ml_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    n_dev = min(_lib.device_count(), 4)
    single = _lib.Context(0)
    grp = _lib.Context.group(n_dev)
    try:
        assert grp.shard_devices == list(range(n_dev))
        assert grp.reduce_kind == "group-rccl" and grp.rccl_ranks == n_dev
        X, pi, mu, S = _mixture(16, 8, 40001, 2)
        g, s = _lib.Data(grp, X), _lib.Data(single, X)
        a, b = g.em_iterate(pi, mu, S, 30, atol=1e-9), s.em_iterate(pi, mu, S, 30, atol=1e-9)
        assert a[0] == b[0] and a[1] == b[1] and np.max(np.abs(a[6] - b[6]) / np.abs(b[6])) < 1e-12
        assert np.array_equal(g.em_labels(8), s.em_labels(8))
        ka, kb = g.kmeans_iterate(X[:8].copy(), 20), s.kmeans_iterate(X[:8].copy(), 20)
        assert ka[0] == kb[0] and relerr(ka[4], kb[4]) < 1e-12 and np.array_equal(g.kmeans_labels(), s.kmeans_labels())
        g.close(); s.close()
    finally:
        grp.close(); single.close()


def test_facade_fit_through_a_group_matches_the_single_gpu_fit():
    """ml::EM::fit / KMeans::fit (through the Python mirror of cppyml.clustering) on a group context: the caller hands over ONE
    N x d array, exactly as with the reference, and gets the single-GPU results."""
    import ctypes as C
    from ml_amd import _lib, cppyml
    from ml_amd.cppyml import clustering
    X, _, mu, _ = _mixture(6, 4, 40000, 21)

    def fits():
        out = {}
        for name, init in (("kpp", clustering.KPP()), ("rp", clustering.RandomPartition()), ("forgy", clustering.Forgy())):
            em = clustering.EM(4)
            em.set_seed(42); em.set_means_initialiser(init); em.set_absolute_tolerance(1e-9); em.set_relative_tolerance(0)
            em.set_maximum_steps(60)
            conv = em.fit(X)
            out["em_" + name] = (conv, em.log_likelihood, np.array(em.means), np.array(em.mixing_probabilities), np.array(em.labels),
                                 np.array(em.responsibilities[:50]))
            km = clustering.KMeans(4)
            km.set_seed(7); km.set_centroids_initialiser(init); km.set_number_initialisations(2)
            conv = km.fit(X)
            out["km_" + name] = (conv, km.inertia, np.array(km.centroids), np.array(km.labels_array))
        em = clustering.EM(4)
        em.set_seed(3); em.set_maximise_first(True)
        em.fit(X)
        out["em_mf"] = (em.log_likelihood, np.array(em.means))
        return out

    ref = fits()
    grp = _lib.Context.group(4, device_ids=[0] * 4)
    try:
        _lib.check(_lib.lib.mlpp_device_set_context(grp.handle))
        got = fits()
    finally:
        _lib.check(_lib.lib.mlpp_device_set_context(None))
        grp.close()
    for key in ref:
        for a, b in zip(ref[key], got[key]):
            if isinstance(a, np.ndarray) and a.dtype.kind in "iu":
                assert np.array_equal(a, b), key
            elif isinstance(a, (bool, np.bool_)):
                assert a == b, key
            else:
                assert relerr(b, a) < 1e-9, key


def test_group_upload_from_device_memory(group3, single):
    """mlhip_data_upload_dev through a group: the block already sits in device memory (a torch tensor here); every shard reads its
    rows from there (same GPU, or a peer with access enabled -- refused, not faulted on, otherwise)."""
    import torch
    from ml_amd import _lib
    X, pi, mu, S = _mixture(5, 3, 10001, 4)
    t = torch.from_numpy(X).to("cuda:0")
    torch.cuda.synchronize()
    g = _lib.Data(group3, device_ptr=t.data_ptr(), shape=X.shape)
    s = _lib.Data(single, X)
    a, b = g.em_step(pi, mu, S), s.em_step(pi, mu, S)
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and relerr(a[2], b[2]) < 1e-11
    assert np.array_equal(g.em_labels(3), s.em_labels(3))
    g.close(); s.close()


def test_group_reduce_selection_and_a_group_of_one(single, monkeypatch):
    """MLHIP_GROUP_REDUCE: `rccl` on shards that share a GPU is refused with the reason (RCCL admits one rank per device), `direct` is
    the in-process sum; a group of ONE shard is an ordinary context behind the group interface (no exchange step at all)."""
    from ml_amd import _lib
    monkeypatch.setenv("MLHIP_GROUP_REDUCE", "rccl")
    with pytest.raises(_lib.MlhipError, match="one GPU"):
        _lib.Context.group(2, device_ids=[0, 0])
    monkeypatch.setenv("MLHIP_GROUP_REDUCE", "nonsense")
    with pytest.raises(ValueError):
        _lib.Context.group(2, device_ids=[0, 0])
    monkeypatch.setenv("MLHIP_GROUP_REDUCE", "direct")
    g2 = _lib.Context.group(2, device_ids=[0, 0])
    assert g2.reduce_kind == "group-direct"
    g2.close()
    monkeypatch.delenv("MLHIP_GROUP_REDUCE")
    one = _lib.Context.group(1, device_ids=[0])
    try:
        assert one.shards == 1 and one.reduce_kind == "none"
        X, pi, mu, S = _mixture(6, 4, 5000, 12)
        a, b = _lib.Data(one, X), _lib.Data(single, X)
        ra, rb = a.em_iterate(pi, mu, S, 8), b.em_iterate(pi, mu, S, 8)
        assert ra[2] == rb[2] and np.array_equal(ra[4], rb[4])          # one shard: the very same arithmetic
        a.close(); b.close()
    finally:
        one.close()
    with pytest.raises(ValueError):
        _lib.Context.group(2, device_ids=[0, 99])                      # no such GPU
