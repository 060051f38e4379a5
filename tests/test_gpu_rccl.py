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
