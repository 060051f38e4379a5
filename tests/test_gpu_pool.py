"""The per-context block cache (runtime/internal.hpp BufferPool): data handles created and released one after the other on one
context get each other's device and pinned blocks back, DIRTY. Nothing may depend on what a fresh allocation happens to contain or
on which block a buffer landed in: every result equals, bit for bit, the one of a context that takes each buffer from the driver
(MLHIP_POOL=0) -- EM (full, diagonal, fused and two-kernel shapes), K-means, the initialisers' weights."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [(3000, 4, 3), (50000, 2, 3), (3000, 4, 3), (20000, 16, 8), (2500, 8, 20), (40000, 32, 5), (3000, 4, 3), (9000, 6, 40), (1200, 3, 2)]


def _run(ctx):
    from ml_amd import _lib
    out = []
    for n, d, K in CASES:
        rng = np.random.default_rng(n + 7 * d + K)
        means = 3.0 * rng.standard_normal((K, d))
        X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.standard_normal((n, d)) + 10.0)
        mu0 = means + 10.0 + 0.2 * rng.standard_normal((K, d))
        dt = _lib.Data(ctx, X)
        _, cov = dt.sample_covariance()
        pi0 = np.full(K, 1.0 / K)
        out.append(dt.em_iterate(pi0, mu0, np.stack([cov] * K), 12, atol=1e-9))
        out.append((dt.em_labels(K),))
        if d <= 32:
            out.append(dt.em_iterate(pi0, mu0, np.stack([np.diag(cov).copy()] * K), 8, diagonal=True))
        out.append(dt.kmeans_iterate(mu0, 15, 0.0))
        out.append((dt.kmeans_labels(), dt.kmeans_distances()))
        out.append(dt.em_step(pi0, mu0, np.stack([cov] * K)))
        out.append((dt.em_responsibilities(K),))
        dt.close()
    return out


def test_results_do_not_depend_on_reused_blocks(monkeypatch):
    from ml_amd import _lib
    monkeypatch.setenv("MLHIP_POOL", "0")
    plain = _lib.Context()
    ref = _run(plain)
    plain.close()
    monkeypatch.delenv("MLHIP_POOL")
    pooled = _lib.Context()
    for _ in range(2):                      # the second round runs entirely on recycled blocks
        got = _run(pooled)
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            for u, v in zip(a, b):
                assert np.array_equal(np.asarray(u), np.asarray(v), equal_nan=True)
    pooled.close()


def test_a_block_may_outlive_its_context():
    """mlhip_ctx_destroy detaches the data handles still alive on the context: freeing one afterwards releases its memory to the driver
    instead of touching the destroyed context's block cache."""
    from ml_amd import _lib
    rng = np.random.default_rng(0)
    ctx = _lib.Context()
    dt = _lib.Data(ctx, rng.standard_normal((5000, 6)))
    dt.sample_covariance()
    ctx._blocks.discard(dt)                 # (the Python wrapper would close the block first)
    ctx.close()
    dt.close()
    again = _lib.Context()                  # the library is still in order
    d2 = _lib.Data(again, rng.standard_normal((300, 3)))
    mean, cov = d2.sample_covariance()
    assert np.all(np.isfinite(cov))
    d2.close()
    again.close()
