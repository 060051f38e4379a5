"""Row-sharded multi-rank path on CPU: world_size 2 over gloo. Covers what runs on the host in an N > 1 job:
shard bounds, the statistics all-reduce hook (ml_amd.dist), and the M-step closing arithmetic applied to the reduced
statistics -- checked against the CPU oracle's M-step on the whole data. The per-shard statistics are produced here
by a numpy restatement of the documented layout (on a GPU box the HIP kernel produces them: tests/test_gpu_*.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_tile_the_rows():
    from ml_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 9, 10_000_000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import torch.distributed as dist
        from ml_amd import _lib
        from ml_amd.dist import allreduce_sum, shard_bounds
        from test_host_facade import finalize, packed_statistics
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        d, K, n = 6, 4, 1001
        rng = np.random.default_rng(7)          # same data on every rank; each takes its shard
        X = 2 + rng.standard_normal((n, d))
        R = rng.dirichlet(np.ones(K), n)
        lo, hi = shard_bounds(n, world, rank)
        # global shift: all-reduce of [column sums, count], exactly what mlhip_data_upload does
        v = np.concatenate([X[lo:hi].sum(axis=0), [hi - lo]])
        allreduce_sum(v.ctypes.data, v.size, False, 0)
        assert v[-1] == n
        shift = v[:-1] / v[-1]
        stats = np.ascontiguousarray(packed_statistics(X[lo:hi], R[lo:hi], shift))
        flat = np.concatenate([stats.ravel(), [float(hi - lo)]])   # [K*F statistics, extra slot] like the device buffer
        allreduce_sum(flat.ctypes.data, flat.size, False, 0)
        assert flat[-1] == n
        pi, mu, S = finalize(d, K, flat[:-1].reshape(K, -1), shift, n)
        q.put((rank, pi, mu, S))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


def test_two_rank_statistics_allreduce_matches_single_rank_oracle(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] is not "error", r[2]
    results.sort(key=lambda r: r[0])
    # every rank ends with bit-identical parameters
    for a, b in zip(results[0][1:], results[1][1:]):
        assert np.array_equal(a, b)
    d, K, n = 6, 4, 1001
    rng = np.random.default_rng(7)
    X = 2 + rng.standard_normal((n, d))
    R = rng.dirichlet(np.ones(K), n)
    em = oracle.EM(K)
    em.set_responsibilities(R, d)
    em.maximisation_step(X)
    _, pi, mu, S = results[0]
    assert np.max(np.abs(pi - em.mixing_probabilities)) < 1e-14
    assert np.max(np.abs(mu - em.means)) < 1e-13 * np.max(np.abs(em.means))
    assert np.max(np.abs(S - em.covariances)) < 1e-12 * np.max(np.abs(em.covariances))


# ---- two ranks on one real GPU (host-side gloo all-reduce): the whole product path, row-sharded ---------------------

def _gpu_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        from ml_amd import _lib, synth
        from ml_amd import dist as mldist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        d, K, n = 8, 5, 6001
        mix = synth.Mixture(d, K, seed=4)
        X, _ = mix.sample(n)
        lo, hi = mldist.shard_bounds(n, world, rank)
        ctx = _lib.Context(0)
        mldist.install_allreduce(ctx, world, rank, on_device=False)
        data = _lib.Data(ctx, np.ascontiguousarray(X[lo:hi]))
        assert data.n_global == n
        mean, cov = data.sample_covariance()
        pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
        lls = []
        for _ in range(3):
            ll, pi, mu, S = data.em_step(pi, mu, S)
            lls.append(ll)
        labels = data.em_labels(K)
        inertia, changed, counts, C1 = data.kmeans_step(mu)
        q.put((rank, lo, hi, data.shift, cov, lls, pi, mu, S, labels, inertia, changed, counts, C1))
        data.close()
        ctx.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_rank():
    """Row-sharded EM / K-means over 2 processes (gloo host hook) == the single-process run on the whole data."""
    import torch.multiprocessing as mp
    from ml_amd import _lib, synth
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] != "error", r[2]
    results.sort(key=lambda r: r[0])
    a, b = results
    # replicated quantities are bit-identical on both ranks
    for i in (3, 4, 6, 7, 8, 10, 12, 13):
        assert np.array_equal(np.asarray(a[i]), np.asarray(b[i])), i
    assert a[5] == b[5]

    d, K, n = 8, 5, 6001
    mix = synth.Mixture(d, K, seed=4)
    X, _ = mix.sample(n)
    ctx = _lib.Context(0)
    data = _lib.Data(ctx, X)
    mean, cov = data.sample_covariance()
    assert np.max(np.abs(a[3] - data.shift)) <= 1e-15 * np.max(np.abs(data.shift)) + 1e-16
    assert np.max(np.abs(a[4] - cov)) <= 1e-13 * np.max(np.abs(cov))
    pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
    for it in range(3):
        ll, pi, mu, S = data.em_step(pi, mu, S)
        assert abs(ll - a[5][it]) <= 1e-12 * abs(ll)
    assert np.max(np.abs(pi - a[6])) <= 1e-12
    assert np.max(np.abs(mu - a[7])) <= 1e-11 * np.max(np.abs(mu))
    assert np.max(np.abs(S - a[8])) <= 1e-10 * np.max(np.abs(S))
    labels = data.em_labels(K)
    assert np.array_equal(labels, np.concatenate([a[9], b[9]]))       # shards tile the rows in order
    inertia, changed, counts, C1 = data.kmeans_step(a[7])
    assert abs(inertia - a[10]) <= 1e-13 * inertia
    assert changed == a[11] == n
    assert np.array_equal(counts, a[12])
    assert np.max(np.abs(C1 - a[13])) <= 1e-14 * np.max(np.abs(C1))
    data.close()
    ctx.close()


# ---- the same through the drop-in classes: every rank calls EM.fit / KMeans.fit on its shard ------------------------

def _facade_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        from ml_amd import cppyml, synth
        from ml_amd import dist as mldist
        from ml_amd.cppyml import clustering as cl
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        mldist.install_allreduce(cppyml.device_context(), world, rank, on_device=False)
        d, K, n = 6, 4, 5003
        mix = synth.Mixture(d, K, seed=8)
        X, _ = mix.sample(n)
        lo, hi = mldist.shard_bounds(n, world, rank)
        shard = np.ascontiguousarray(X[lo:hi])
        out = {}
        for name, init in (("fixed", cl.FixedCentroids(mix.initial_means())), ("forgy", cl.Forgy()), ("kpp", cl.KPP()),
                           ("random_partition", cl.RandomPartition())):
            em = cl.EM(K)
            em.set_means_initialiser(init)
            em.set_absolute_tolerance(1e-10)
            em.set_relative_tolerance(1e-10)
            em.set_maximum_steps(200)
            em.set_seed(5)
            conv = em.fit(shard)
            out[name] = (conv, em.steps_done, em.log_likelihood, em.means.copy(), em.mixing_probabilities.copy(),
                         np.stack([em.covariance(k) for k in range(K)]), np.asarray(em.labels))
        km = cl.KMeans(K)
        km.set_centroids_initialiser(cl.FixedCentroids(mix.initial_means()))
        conv = km.fit(shard)
        out["kmeans"] = (conv, km.steps_done, km.inertia, km.centroids.copy(), np.asarray(km.labels_array))
        km = cl.KMeans(K)
        km.set_centroids_initialiser(cl.KPP())
        km.set_seed(11)
        km.set_number_initialisations(3)
        conv = km.fit(shard)
        out["kmeans_kpp"] = (conv, km.steps_done, km.inertia, km.centroids.copy(), np.asarray(km.labels_array))
        q.put((rank, lo, hi, out))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


@pytest.mark.gpu
def test_facade_fit_row_sharded_over_two_ranks():
    """cppyml.clustering.EM / KMeans fitted by 2 processes on row shards (hook installed on the facade's context) ==
    the single-process fit on the whole data; rank 0's initial means are what every rank starts from."""
    import torch.multiprocessing as mp
    from ml_amd import synth
    from ml_amd.cppyml import clustering as cl
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_facade_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] != "error", r[2]
    results.sort(key=lambda r: r[0])
    a, b = results[0][3], results[1][3]
    for name in ("fixed", "forgy", "kpp", "random_partition"):
        assert a[name][0] == b[name][0] and a[name][1] == b[name][1]
        for i in (2, 3, 4, 5):                                           # replicated: bit-identical on both ranks
            assert np.array_equal(np.asarray(a[name][i]), np.asarray(b[name][i])), (name, i)
    assert a["kmeans"][0] == b["kmeans"][0] and a["kmeans"][1] == b["kmeans"][1] and a["kmeans"][2] == b["kmeans"][2]
    assert np.array_equal(a["kmeans"][3], b["kmeans"][3])

    d, K, n = 6, 4, 5003
    mix = synth.Mixture(d, K, seed=8)
    X, _ = mix.sample(n)
    # the library initialisers draw, on the shards, exactly what one process draws on the whole sample (same seed)
    for name, init in (("fixed", cl.FixedCentroids(mix.initial_means())), ("forgy", cl.Forgy()), ("kpp", cl.KPP()),
                       ("random_partition", cl.RandomPartition())):
        em = cl.EM(K)
        em.set_means_initialiser(init)
        em.set_absolute_tolerance(1e-10)
        em.set_relative_tolerance(1e-10)
        em.set_maximum_steps(200)
        em.set_seed(5)
        conv1 = em.fit(X)
        conv, steps, ll, means, pis, covs, _ = a[name]
        assert conv == conv1 and steps == em.steps_done, name
        assert abs(ll - em.log_likelihood) <= 1e-12 * abs(ll), name
        assert np.max(np.abs(means - em.means)) <= 1e-10 * np.max(np.abs(means)), name
        assert np.max(np.abs(pis - em.mixing_probabilities)) <= 1e-11, name
        for k in range(K):
            assert np.max(np.abs(covs[k] - em.covariance(k))) <= 1e-9 * np.max(np.abs(covs[k])), name
        if conv:
            assert np.array_equal(np.concatenate([a[name][6], b[name][6]]), np.asarray(em.labels)), name
    km = cl.KMeans(K)
    km.set_centroids_initialiser(cl.KPP())
    km.set_seed(11)
    km.set_number_initialisations(3)
    assert km.fit(X) == a["kmeans_kpp"][0]
    assert km.steps_done == a["kmeans_kpp"][1]
    assert abs(km.inertia - a["kmeans_kpp"][2]) <= 1e-13 * km.inertia
    assert np.max(np.abs(km.centroids - a["kmeans_kpp"][3])) <= 1e-13 * np.max(np.abs(km.centroids))
    assert np.array_equal(np.concatenate([a["kmeans_kpp"][4], b["kmeans_kpp"][4]]), np.asarray(km.labels_array))
    km = cl.KMeans(K)
    km.set_centroids_initialiser(cl.FixedCentroids(mix.initial_means()))
    assert km.fit(X) == a["kmeans"][0]
    assert km.steps_done == a["kmeans"][1]
    assert abs(km.inertia - a["kmeans"][2]) <= 1e-13 * km.inertia
    assert np.max(np.abs(km.centroids - a["kmeans"][3])) <= 1e-13 * np.max(np.abs(km.centroids))
    assert np.array_equal(np.concatenate([a["kmeans"][4], b["kmeans"][4]]), np.asarray(km.labels_array))


# ---- shards with <= K rows: decisions are taken from the GLOBAL sample size on every rank -----------------

def _small_shard_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        from ml_amd import cppyml
        from ml_amd import dist as mldist
        from ml_amd.cppyml import clustering as cl
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        mldist.install_allreduce(cppyml.device_context(), world, rank, on_device=False)
        d, K = 3, 4
        rng = np.random.default_rng(3)
        centres = 8.0 * rng.standard_normal((K, d))
        X = np.ascontiguousarray(centres[np.arange(60) % K] + 0.3 * rng.standard_normal((60, d)))
        out = {}
        # (a) uneven shards: rank 1 holds exactly K rows (it used to take the local "exact fit" shortcut), then fewer than
        #     K, then none at all
        for name, cut in (("k_rows", 60 - K), ("two_rows", 58), ("empty", 60)):
            shard = np.ascontiguousarray(X[:cut] if rank == 0 else X[cut:])
            em = cl.EM(K)
            em.set_means_initialiser(cl.FixedCentroids(centres + 0.1))
            em.set_maximum_steps(50)
            conv = em.fit(shard)
            km = cl.KMeans(K)
            km.set_centroids_initialiser(cl.FixedCentroids(centres + 0.1))
            kconv = km.fit(shard)
            out[name] = (conv, em.steps_done, em.log_likelihood, em.means.copy(), np.asarray(em.labels),
                         kconv, km.inertia, km.centroids.copy(), np.asarray(km.labels_array))
        # (b) the whole sample has exactly K rows, 3 + 1 over the ranks: the exact fit, put together across ranks
        shard = np.ascontiguousarray(X[:3] if rank == 0 else X[3:4])
        em = cl.EM(K)
        conv = em.fit(shard)
        km = cl.KMeans(K)
        kconv = km.fit(shard)
        out["exact"] = (conv, em.log_likelihood, em.means.copy(), np.asarray(em.labels), em.responsibilities.copy(),
                        kconv, km.inertia, km.centroids.copy(), np.asarray(km.labels_array))
        # (c) fewer than K rows in total: every rank raises, nobody is left waiting in a collective
        shard = np.ascontiguousarray(X[:2] if rank == 0 else X[2:3])
        errors = []
        for model in (cl.EM(K), cl.KMeans(K)):
            try:
                model.fit(shard)
                errors.append(None)
            except ValueError as e:
                errors.append(str(e))
        out["too_few"] = errors
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


@pytest.mark.gpu
def test_small_and_empty_shards_follow_the_global_sample_size():
    import torch.multiprocessing as mp
    from ml_amd.cppyml import clustering as cl
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_small_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] != "error", r[2]
    results.sort(key=lambda r: r[0])
    a, b = results[0][1], results[1][1]

    d, K = 3, 4
    rng = np.random.default_rng(3)
    centres = 8.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(centres[np.arange(60) % K] + 0.3 * rng.standard_normal((60, d)))
    em = cl.EM(K)
    em.set_means_initialiser(cl.FixedCentroids(centres + 0.1))
    em.set_maximum_steps(50)
    conv1 = em.fit(X)
    km = cl.KMeans(K)
    km.set_centroids_initialiser(cl.FixedCentroids(centres + 0.1))
    kconv1 = km.fit(X)
    for name in ("k_rows", "two_rows", "empty"):
        ra, rb = a[name], b[name]
        assert ra[0] == rb[0] == conv1 and ra[1] == rb[1] == em.steps_done, name
        assert ra[2] == rb[2] and abs(ra[2] - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood), name
        assert np.array_equal(ra[3], rb[3]) and np.max(np.abs(ra[3] - em.means)) <= 1e-11 * np.max(np.abs(em.means)), name
        assert np.array_equal(np.concatenate([ra[4], rb[4]]), np.asarray(em.labels)), name
        assert ra[5] == rb[5] == kconv1 and abs(ra[6] - km.inertia) <= 1e-13 * km.inertia, name
        assert np.array_equal(ra[7], rb[7]) and np.max(np.abs(ra[7] - km.centroids)) <= 1e-13 * np.max(np.abs(km.centroids)), name
        assert np.array_equal(np.concatenate([ra[8], rb[8]]), np.asarray(km.labels_array)), name

    ea, eb = a["exact"], b["exact"]
    assert ea[0] and eb[0] and ea[1] == eb[1] == np.inf
    assert np.array_equal(ea[2], eb[2]) and np.array_equal(ea[2], X[:4].T)          # means d x K = the samples themselves
    assert list(ea[3]) == [0, 1, 2] and list(eb[3]) == [3]
    assert np.array_equal(ea[4], np.eye(4)[:3]) and np.array_equal(eb[4], np.eye(4)[3:])
    assert ea[5] and eb[5] and ea[6] == eb[6] == 0
    assert np.array_equal(ea[7], eb[7]) and np.array_equal(ea[7], X[:4])             # centroids K x d
    assert list(ea[8]) == [0, 1, 2] and list(eb[8]) == [3]

    for errs in (a["too_few"], b["too_few"]):
        assert all(e and "Not enough data" in e for e in errs), errs


# ---- a rank that receives different sums must not go unnoticed -------------------------------------------------------

def _diverging_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        from ml_amd import _lib, synth
        from ml_amd import dist as mldist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        d, K, n = 8, 5, 4001
        mix = synth.Mixture(d, K, seed=4)
        X, _ = mix.sample(n)
        lo, hi = mldist.shard_bounds(n, world, rank)
        ctx = _lib.Context(0)

        def hook(ptr, count, on_device, stream):
            mldist.allreduce_sum(ptr, count, on_device, stream)
            if rank == 1 and count > 16:            # the statistics buffers, not the small exchanges (shift, checksums)
                import ctypes
                buf = (ctypes.c_double * count).from_address(ptr)
                buf[0] *= 1.0 + 1e-12               # what a collective that is not bitwise reproducible across ranks would do

        ctx.set_allreduce(hook, False, world, rank)
        data = _lib.Data(ctx, np.ascontiguousarray(X[lo:hi]))
        _, cov = data.sample_covariance()
        pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
        msgs = []
        for call in (lambda: data.em_iterate(pi, mu, S, 3), lambda: data.kmeans_iterate(mu, 3)):
            try:
                call()
                msgs.append(None)
            except _lib.MlhipError as e:
                msgs.append(str(e))
        q.put((rank, msgs))
        data.close()
        ctx.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


@pytest.mark.gpu
def test_diverging_ranks_fail_the_fit_on_every_rank():
    """Parameters are never broadcast (every rank closes the same all-reduced sums): the end-of-fit checksum exchange of
    mlhip_em_iterate / mlhip_kmeans_iterate must catch a rank whose sums differ in the last bits."""
    import torch.multiprocessing as mp
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_diverging_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] != "error", r[2]
        assert len(r[1]) == 2 and all(m and "ranks disagree" in m for m in r[1]), r


def _kpp_worker(rank, world, port, q, scale):
    try:
        sys.path.insert(0, ROOT)
        if scale:
            os.environ["MLHIP_KPP_DELTA_SCALE"] = scale
        import torch.distributed as dist
        from ml_amd import cppyml, synth
        from ml_amd import dist as mldist
        from ml_amd.cppyml import clustering as cl
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        mldist.install_allreduce(cppyml.device_context(), world, rank, on_device=False)
        d, K, n = 5, 12, 120001
        X, _ = synth.Mixture(d, K, seed=8).sample(n)
        X[1000:1040] = X[7]
        lo, hi = mldist.shard_bounds(n, world, rank)
        if world == 3:                                     # uneven shards, the last one small
            lo, hi = ((0, 70000), (70000, 119990), (119990, n))[rank]
        km = cl.KMeans(K)
        km.set_centroids_initialiser(cl.KPP())
        km.set_seed(123)
        km.set_maximum_steps(2)
        km.fit(np.ascontiguousarray(X[lo:hi]))
        q.put((rank, km.centroids.copy()))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale", [(2, ""), (3, ""), (2, "1e9")])
def test_sharded_kpp_draws_on_the_devices_match_the_single_process_draws(oracle, world, scale):
    """N = 120 001 over 2 / 3 row shards: K-means++ takes its draws on the devices (mlhip_kpp_draw with the ranks' weight sums and
    candidates exchanged through the all-reduce hook) and picks the rows the oracle's sequential std::discrete_distribution picks on the
    whole sample -- also when the bound is widened until every draw goes back to the sequential evaluation over the ranks."""
    import torch.multiprocessing as mp
    from ml_amd import synth
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_kpp_worker, args=(r, world, port, q, scale)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] is not None and not (isinstance(r[1], str) and r[1] == "error"), r[2]
    d, K, n = 5, 12, 120001
    X, _ = synth.Mixture(d, K, seed=8).sample(n)
    X[1000:1040] = X[7]
    okm = oracle.KMeans(K)
    okm.set_centroids_initialiser(oracle.KPP)
    okm.set_seed(123)
    okm.set_maximum_steps(2)
    okm.fit(X)
    for r in results:
        assert np.max(np.abs(r[1] - okm.centroids)) <= 1e-13 * np.max(np.abs(okm.centroids)), r[0]
