"""The device-resident K-means step loop (kmeans_resident.hip: the whole loop of KMeans::fit_once, ML/KMeans.cpp:80-110, in ONE launch of
one workgroup for small blocks -- the reference's own benchmark sizes, Benchmarks/bm_KMeans.cpp) against the three-launch loop of
runtime/kmeans.cpp (MLHIP_RESIDENT=0), which tests/test_gpu_iterate.py and test_gpu_facade.py pin to the oracle: BIT-identical steps,
convergence flag, inertia, counts, centroids, previous centroids, labels and distances -- the kernel evaluates the same distances, the
same exact sums, the same closing arithmetic and adds the inertia in the launches' order. Also against the oracle's loop directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


def _problem(d, K, n, seed, spread=3.0):
    rng = np.random.default_rng(seed)
    means = spread * rng.standard_normal((K, d))
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)))
    c0 = X[rng.choice(n, K, replace=False)].copy()
    return X, c0


def _same(a, b):
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (a[:3], b[:3])
    for x, y in zip(a[3:], b[3:]):
        assert np.array_equal(x, y)


SHAPES = [(2, 3, 4096), (2, 3, 100), (2, 3, 1000), (2, 3, 3000), (1, 16, 4000), (1, 1, 300), (2, 10, 4096), (3, 6, 2048),
          (6, 2, 1000), (5, 7, 1024), (3, 32, 2000), (2, 5, 65), (6, 32, 1023)]


@pytest.mark.parametrize("d,K,n", SHAPES)
def test_resident_loop_is_bit_identical_to_the_three_launch_loop(ctx, d, K, n, monkeypatch):
    from ml_amd import _lib
    X, c0 = _problem(d, K, n, 31 * d + K)
    dt = _lib.Data(ctx, X)
    runs = [(60, 1e-12), (4, 0.0), (1, 0.0), (60, 1e-2)]          # identical labels / max_steps / one step / the tolerance test
    monkeypatch.setenv("MLHIP_RESIDENT", "0")
    ref = []
    for steps, atol in runs:
        r = dt.kmeans_iterate(c0, steps, atol)
        ref.append((r, dt.kmeans_labels(), dt.kmeans_distances()))
    monkeypatch.delenv("MLHIP_RESIDENT")
    ctx.timing_enable(True)
    for (steps, atol), (r, lab, dist) in zip(runs, ref):
        ctx.timing_reset()
        got = dt.kmeans_iterate(c0, steps, atol)
        assert ctx.timing_get("kmeans_resident")[1] == 1 and ctx.timing_get("kmeans_assign")[1] == 0     # the one launch
        _same(got, r)
        assert np.array_equal(dt.kmeans_labels(), lab)
        assert np.array_equal(dt.kmeans_distances(), dist)
    ctx.timing_enable(False)
    # the state it leaves serves the per-step entry points: a step from the final centroids is the same either way
    fin = ref[0][0][4]
    dt.kmeans_iterate(c0, 60, 1e-12)
    a = dt.kmeans_step(fin)
    monkeypatch.setenv("MLHIP_RESIDENT", "0")
    dt.kmeans_iterate(c0, 60, 1e-12)
    b = dt.kmeans_step(fin)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    dt.close()


@pytest.mark.parametrize("d,K,n", [(2, 3, 4000), (3, 5, 2000), (1, 4, 777)])
def test_resident_loop_against_the_oracle(ctx, oracle, d, K, n):
    """The oracle's own KMeans::fit from the same start (ML/KMeans.cpp:80-110): same number of steps, same decision, same labels."""
    from ml_amd import _lib
    X, c0 = _problem(d, K, n, 5 * d + K)
    dt = _lib.Data(ctx, X)
    ctx.timing_enable(True)
    ctx.timing_reset()
    steps, conv, inertia, counts, cur, old = dt.kmeans_iterate(c0, 100, 1e-14)
    assert ctx.timing_get("kmeans_resident")[1] == 1
    ctx.timing_enable(False)
    km = oracle.KMeans(K)
    km.set_absolute_tolerance(1e-14)
    km.set_maximum_steps(100)
    km.set_centroids_initialiser(oracle.FIXED, c0)
    assert km.fit(X) == conv
    assert km.steps_done == steps
    assert np.array_equal(km.labels, dt.kmeans_labels())
    np.testing.assert_allclose(cur, km.centroids, rtol=0, atol=1e-12)
    dt.close()


def test_empty_cluster_goes_to_the_origin(ctx):
    """ML/KMeans.cpp:184: a cluster that loses all its points moves to the origin -- in both loops."""
    from ml_amd import _lib
    rng = np.random.default_rng(3)
    X = np.ascontiguousarray(rng.standard_normal((500, 2)) + 5.0)
    c0 = np.array([[5.0, 5.0], [4.5, 5.5], [-100.0, -100.0]])
    dt = _lib.Data(ctx, X)
    got = dt.kmeans_iterate(c0, 1, 0.0)
    import os
    os.environ["MLHIP_RESIDENT"] = "0"
    try:
        ref = dt.kmeans_iterate(c0, 1, 0.0)
    finally:
        del os.environ["MLHIP_RESIDENT"]
    _same(got, ref)
    assert got[3][2] == 0.0 and np.all(got[4][2] == 0.0)
    dt.close()


@pytest.mark.parametrize("d,K,n", [(2, 3, 4097), (3, 3, 2049), (6, 3, 1025), (4, 3, 500), (2, 33, 500)])
def test_shapes_beyond_the_one_workgroup_take_the_launches(ctx, d, K, n):
    from ml_amd import _lib
    X, c0 = _problem(d, K, n, 3 * d + K)
    dt = _lib.Data(ctx, X)
    ctx.timing_enable(True)
    ctx.timing_reset()
    dt.kmeans_iterate(c0, 5, 0.0)
    assert ctx.timing_get("kmeans_resident")[1] == 0 and ctx.timing_get("kmeans_assign")[1] >= 1
    ctx.timing_enable(False)
    dt.close()
