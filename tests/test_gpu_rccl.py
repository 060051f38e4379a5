"""Native RCCL in the C library (mlhip_ctx_init_rccl): the statistics all-reduce is ncclAllReduce on the context's own
communicator. A one-GPU box can only hold a 1-rank communicator (RCCL refuses two ranks on one device), which still
exercises the whole route: dlopen of librccl, unique id, ncclCommInitRank, the device-pointer all-reduce on the
context's stream inside every EM / K-means step, ncclCommCount, teardown. The multi-rank arithmetic around it is
covered by the gloo tests (tests/test_dist_gloo.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fit_steps(data, mix, K, steps=3):
    _, cov = data.sample_covariance()
    pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
    lls = []
    for _ in range(steps):
        ll, pi, mu, S = data.em_step(pi, mu, S)
        lls.append(ll)
    return lls, pi, mu, S, data.em_labels(K), data.kmeans_step(mu)


@pytest.mark.parametrize("rendezvous", ["bytes", "file"])
def test_one_rank_communicator_runs_every_step_through_rccl(tmp_path, rendezvous):
    from ml_amd import _lib, synth
    d, K, n = 16, 6, 20000
    mix = synth.Mixture(d, K, seed=21)
    X, _ = mix.sample(n)

    plain = _lib.Context(0)
    ref = _fit_steps(_lib.Data(plain, X), mix, K)
    plain.close()

    ctx = _lib.Context(0)
    assert ctx.rccl_ranks == 0 and ctx.world == (1, 0)
    if rendezvous == "bytes":
        uid = _lib.rccl_unique_id()
        assert len(uid) == _lib.RCCL_UNIQUE_ID_BYTES
        ctx.init_rccl(uid, 1, 0)
    else:
        ctx.init_rccl_file(os.path.join(tmp_path, "rccl_id"), 1, 0)
    assert ctx.rccl_ranks == 1 and ctx.world == (1, 0)
    data = _lib.Data(ctx, X)                       # the upload's shift all-reduce goes through the communicator too
    assert data.n_global == n
    got = _fit_steps(data, mix, K)
    # a sum over one rank is the identity: bit-identical to the run without a communicator
    assert got[0] == ref[0]
    for a, b in zip(got[1:5], ref[1:5]):
        assert np.array_equal(a, b)
    assert got[5][0] == ref[5][0] and got[5][1] == ref[5][1]
    assert np.array_equal(got[5][2], ref[5][2]) and np.array_equal(got[5][3], ref[5][3])
    ctx.finalize_rccl()
    assert ctx.rccl_ranks == 0 and ctx.world == (1, 0)
    data.close()
    ctx.close()


def test_bad_arguments_are_refused():
    from ml_amd import _lib
    ctx = _lib.Context(0)
    with pytest.raises(ValueError):
        ctx.init_rccl(b"short", 1, 0)
    with pytest.raises(ValueError):
        ctx.init_rccl(_lib.rccl_unique_id(), 2, 2)           # rank out of range
    ctx.close()


# ---- more than one rank: needs as many GPUs (the driver's 8-GPU node; skipped on a one-GPU box) ------------------------

def _rccl_worker(rank, world, path, q):
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ml_amd import _lib, synth
        from ml_amd.dist import shard_bounds
        d, K, n = 16, 6, 40001
        mix = synth.Mixture(d, K, seed=21)
        X, _ = mix.sample(n)
        lo, hi = shard_bounds(n, world, rank)
        ctx = _lib.Context(rank)                              # one process per GPU
        ctx.init_rccl_file(path, world, rank)                 # no torch, no Python in the iterations
        assert ctx.rccl_ranks == world and ctx.world == (world, rank)
        data = _lib.Data(ctx, np.ascontiguousarray(X[lo:hi]))
        assert data.n_global == n
        _, cov = data.sample_covariance()
        pi0, mu0, S0 = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
        steps, conv, ll, pi, mu, S, hist = data.em_iterate(pi0, mu0, S0, 50, 1e-10, 1e-10)     # ends in the rank-consistency check
        labels = data.em_labels(K)
        ksteps, kconv, inertia, counts, C, _ = data.kmeans_iterate(mu0, 30)
        q.put((rank, lo, hi, steps, conv, ll, pi, mu, S, labels, ksteps, kconv, inertia, counts, C, data.kmeans_labels()))
        data.close()
        ctx.finalize_rccl()
        ctx.close()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


@pytest.mark.parametrize("world", [2, 4])      # (+ the parent: at most 5 processes on the GPUs at once)
def test_native_rccl_ranks_reproduce_the_single_gpu_fit(tmp_path, world):
    """mlhip_ctx_init_rccl_file with one process per GPU: the row-sharded EM and K-means fits (statistics all-reduced by
    ncclAllReduce on the library's own communicators) against the single-GPU fit of the whole sample: same steps, parameters to
    tolerance (the summation order differs), labels bit-exact; every rank holds bit-identical parameters."""
    import torch.multiprocessing as mp
    from ml_amd import _lib, synth
    if _lib.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_lib.device_count()}")
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    path = os.path.join(tmp_path, "rccl_id")
    procs = [mpctx.Process(target=_rccl_worker, args=(r, world, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
            p.join()
    for r in results:
        assert r[1] != "error", r[2]
    results.sort(key=lambda r: r[0])
    assert not os.path.exists(path)                      # rank 0 removed the rendezvous file
    first = results[0]
    for r in results[1:]:
        for i in (3, 4, 5, 10, 11, 12):
            assert r[i] == first[i], i
        for i in (6, 7, 8, 13, 14):
            assert np.array_equal(r[i], first[i]), i     # replicated quantities: bit-identical on every rank

    d, K, n = 16, 6, 40001
    mix = synth.Mixture(d, K, seed=21)
    X, _ = mix.sample(n)
    ctx = _lib.Context(0)
    data = _lib.Data(ctx, X)
    _, cov = data.sample_covariance()
    steps, conv, ll, pi, mu, S, hist = data.em_iterate(np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K), 50, 1e-10, 1e-10)
    assert (first[3], first[4]) == (steps, conv) and abs(first[5] - ll) <= 1e-12 * abs(ll)
    assert np.max(np.abs(first[7] - mu)) <= 1e-10 * np.max(np.abs(mu)) and np.max(np.abs(first[8] - S)) <= 1e-9 * np.max(np.abs(S))
    assert np.array_equal(np.concatenate([r[9] for r in results]), data.em_labels(K))
    ksteps, kconv, inertia, counts, C, _ = data.kmeans_iterate(mix.initial_means(), 30)
    assert (first[10], first[11]) == (ksteps, kconv) and abs(first[12] - inertia) <= 1e-13 * inertia
    assert np.array_equal(first[13], counts) and np.max(np.abs(first[14] - C)) <= 1e-13 * np.max(np.abs(C))
    assert np.array_equal(np.concatenate([r[15] for r in results]), data.kmeans_labels())
    data.close()
    ctx.close()
