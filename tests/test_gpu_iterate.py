"""mlhip_em_iterate -- the loop of EM::fit (ML/EM.cpp:143-170) in one call with the M-step's closing arithmetic and the K
covariance factorizations on the device (em_close.hip) -- against the same loop made of mlhip_em_step calls (closing on the
host), which the parity tests pin to the oracle. Same number of steps, log-likelihoods 1e-13, parameters 1e-12, labels
bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def ctx():
    from ml_amd import _lib
    c = _lib.Context()
    yield c
    c.close()


def _problem(d, K, n, seed, spread=3.0, sigma=None):
    rng = np.random.default_rng(seed)
    means = spread * rng.standard_normal((K, d))
    sig = np.ones(K) if sigma is None else np.asarray(sigma, dtype=float)
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(means[comp] + rng.standard_normal((n, d)) * sig[comp][:, None])
    mu0 = means + 0.2 * sig[:, None] * rng.standard_normal((K, d))
    return X, mu0, means, sig


def _step_loop(dt, pi, mu, S, max_steps, atol, rtol, diagonal):
    lls, old = [], None
    conv = False
    for step in range(max_steps):
        fn = dt.em_step_diag if diagonal else dt.em_step
        ll, pi, mu, S = fn(pi, mu, S)
        lls.append(ll)
        if step > 0 and abs(ll - old) < atol + rtol * max(abs(old), abs(ll)):
            conv = True
            break
        old = ll
    return len(lls), conv, lls, pi, mu, S


@pytest.mark.parametrize("d,K,n,diagonal", [
    (32, 64, 30000, False),     # matrix-core E-step (FOLD), self-normalising statistics kernel
    (16, 40, 20000, False),     # matrix-core E-step with its own log-sum-exp (K spans two row-block groups)
    (12, 3, 9000, False),
    (8, 5, 9000, False),        # scalar-fed E-step + wide statistics kernel
    (4, 3, 20000, False),       # fused small-shape kernel
    (4, 3, 9000, False),        # ... with at most 64 partial blocks: their reduction is folded into the closing kernel
    (2, 8, 12000, False),
    (6, 5, 700, False),
    (2, 8, 20000, False),
    (48, 9, 6000, False),
    (64, 4, 5000, False),
    (100, 3, 4000, False),      # d > 64: the loop falls back to the per-step functions
    (16, 16, 30000, True),      # diagonal covariances
    (5, 17, 9000, True),
    (32, 64, 12000, True),
])
def test_iterate_equals_the_step_loop(ctx, oracle, d, K, n, diagonal):
    from ml_amd import _lib
    X, mu0, _, _ = _problem(d, K, n, 100 * d + K)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    S0 = np.stack([np.diag(cov)] * K) if diagonal else np.stack([cov] * K)
    pi0 = np.full(K, 1.0 / K)
    steps_a, conv_a, lls, pi_a, mu_a, S_a = _step_loop(dt, pi0, mu0, S0, 12, 1e-9, 1e-9, diagonal)
    labels_a, R_a = dt.em_labels(K), dt.em_responsibilities(K)
    steps_b, conv_b, ll_b, pi_b, mu_b, S_b, hist = dt.em_iterate(pi0, mu0, S0, 12, 1e-9, 1e-9, diagonal)
    assert steps_b == steps_a and conv_b == conv_a
    assert np.max(np.abs(hist - np.array(lls)) / np.abs(np.array(lls))) < 1e-13
    assert ll_b == hist[-1]
    assert relerr(pi_b, pi_a) < 1e-12 and relerr(mu_b, mu_a) < 1e-12 and relerr(S_b, S_a) < 1e-11
    # ... and against the ORACLE's loop (ML/EM.cpp:143-170 spelt out with expectation_step / maximisation_step): the same
    # trajectory, not only the same as the per-step HIP calls
    ref = oracle.EM(K)
    if diagonal:
        ref.set_covariance_type("diag")
    ref.set_parameters(mu0, np.stack([np.diag(v) for v in S0]) if diagonal else S0, pi0)
    ref_lls, old = [], None
    for step in range(12):
        ref.expectation_step(X)
        ref.maximisation_step(X)
        ref_lls.append(ref.log_likelihood)
        if step > 0 and abs(ref_lls[-1] - old) < 1e-9 + 1e-9 * max(abs(old), abs(ref_lls[-1])):
            break
        old = ref_lls[-1]
    assert len(ref_lls) == steps_b
    assert np.max(np.abs(hist - np.array(ref_lls)) / np.abs(np.array(ref_lls))) < 1e-12
    S_ref = np.stack([np.diag(c) for c in ref.covariances]) if diagonal else ref.covariances
    assert relerr(pi_b, ref.mixing_probabilities) < 1e-11 and relerr(mu_b, ref.means) < 1e-10 and relerr(S_b, S_ref) < 1e-9
    ref.calculate_labels()
    assert np.array_equal(dt.em_labels(K), np.asarray(ref.labels))
    # the E-step results left on the device are those of the LAST iteration, as after the step loop
    assert np.array_equal(dt.em_labels(K), labels_a)
    assert np.max(np.abs(dt.em_responsibilities(K) - R_a)) < 1e-12
    # tolerances 0: exactly max_steps iterations (ML/EM.cpp:163: `ll_change < 0` is never true)
    steps_c, conv_c, *_ = dt.em_iterate(pi0, mu0, S0, 5, 0.0, 0.0, diagonal)
    assert steps_c == 5 and not conv_c
    dt.close()


@pytest.mark.parametrize("d", [12, 8, 4])     # 12: matrix-core E-step (synchronous loop); 8, 4: the lagged loop ROLLS BACK to the flagged iteration
@pytest.mark.parametrize("diagonal", [False, True])
def test_iterate_with_a_far_tight_cluster_takes_the_refinement_route(ctx, oracle, diagonal, d):
    """A component 1000 sigma away from the data mean is flagged by the device closing; that iteration is closed on the host
    with the refinement pass, and the loop goes on. Checked against the oracle's two-pass arithmetic."""
    from ml_amd import _lib
    K, n = 3, 24000
    rng = np.random.default_rng(11)
    centres = np.array([[0.0] * d, [300.0] * d, [-200.0] * d])
    sig = np.array([1.0, 1e-3, 1e-2])
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(centres[comp] + rng.standard_normal((n, d)) * sig[comp][:, None])
    mu0 = centres + 0.1 * sig[:, None] * rng.standard_normal((K, d))
    var0 = np.repeat((sig ** 2)[:, None], d, axis=1) * 1.5
    S0 = var0 if diagonal else np.stack([np.diag(v) for v in var0])
    pi0 = np.full(K, 1.0 / K)
    dt = _lib.Data(ctx, X)
    steps, conv, ll, pi, mu, S, hist = dt.em_iterate(pi0, mu0, S0, 3, 0.0, 0.0, diagonal)
    em = oracle.EM(K)
    if diagonal:
        em.set_covariance_type("diag")
    em.set_parameters(mu0, np.stack([np.diag(v) for v in var0]), pi0)
    for it in range(3):
        em.expectation_step(X)
        assert abs(hist[it] - em.log_likelihood) <= 1e-11 * abs(em.log_likelihood)
        em.maximisation_step(X)
    So = em.covariances
    So = np.stack([np.diag(So[k]) for k in range(K)]) if diagonal else So
    assert relerr(pi, em.mixing_probabilities) < 1e-12
    assert np.max(np.abs(mu - em.means) / np.maximum(1.0, np.abs(em.means))) < 1e-13
    for k in range(K):
        assert relerr(S[k], So[k]) < 1e-8, k
    dt.close()


def test_device_close_can_be_switched_off(ctx, monkeypatch):
    from ml_amd import _lib
    X, mu0, _, _ = _problem(16, 6, 8000, 5)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    S0, pi0 = np.stack([cov] * 6), np.full(6, 1 / 6)
    a = dt.em_iterate(pi0, mu0, S0, 6)
    dt.close()
    # (the switch is read once per process: exercised in a subprocess)
    import subprocess, sys, json, os
    code = (
        "import numpy as np, json, sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))\n"
        "from ml_amd import _lib\n"
        "import test_gpu_iterate as t\n"
        "X, mu0, _, _ = t._problem(16, 6, 8000, 5)\n"
        "ctx = _lib.Context(); dt = _lib.Data(ctx, X); _, cov = dt.sample_covariance()\n"
        "r = dt.em_iterate(np.full(6, 1 / 6), mu0, np.stack([cov] * 6), 6)\n"
        "print(json.dumps([r[0], r[2], r[4].tolist()]))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MLHIP_DEVICE_CLOSE="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    steps, ll, mu = json.loads(out.stdout.strip().splitlines()[-1])
    assert steps == a[0] and abs(ll - a[2]) <= 1e-13 * abs(ll)
    assert relerr(np.array(mu), a[4]) < 1e-12


def _kmeans_step_loop(dt, C, max_steps, atol):
    """The facade's loop (ML/KMeans.cpp:80-110) made of mlhip_kmeans_step / mlhip_kmeans_assign calls."""
    cur, old = np.array(C), np.zeros_like(C)
    steps, conv, inertia, counts = 0, False, None, None
    for step in range(max_steps):
        inertia, changed, counts, upd = dt.kmeans_step(cur)
        steps += 1
        if step > 0 and changed == 0:
            conv = True
            break
        old, cur = cur, upd
        if step > 0:
            shift = 0.0
            for delta in (cur - old).ravel():
                shift += delta * delta
            if shift < atol:
                inertia, _ = dt.kmeans_assign(cur)
                conv = True
                break
    return steps, conv, inertia, counts, cur, old


@pytest.mark.parametrize("d,K,n,max_steps,atol", [
    (8, 256, 40000, 12, 0.0),       # quad-tracking kernel, stops on the step count
    (8, 16, 6000, 200, 0.0),        # runs until the labels repeat
    (4, 5, 3000, 200, 1e-6),        # stops on the centroid shift, then assigns once more
    (2, 7, 5000, 300, 1e-10),       # direct-form kernel
    (6, 130, 9000, 10, 0.0),        # zero-padded copy of the block (d = 6 -> 8 rows)
    (32, 40, 4000, 100, 1e-8),
    (16, 3, 500, 2, 0.0),
    (3, 4, 60, 1, 0.0),             # a single step
    (8, 300, 320, 50, 0.0),         # nearly as many clusters as samples: empty clusters go to the origin
])
def test_kmeans_iterate_equals_the_step_loop(ctx, oracle, d, K, n, max_steps, atol):
    """mlhip_kmeans_iterate -- the step loop of KMeans::fit_once with the centroid table kept on the device between the
    stopping tests -- gives the same step count, decisions, centroids, counts, inertia and labels, bit for bit, as the loop
    of mlhip_kmeans_step calls (closing arithmetic on the host), which the parity tests pin to the oracle."""
    from ml_amd import _lib
    rng = np.random.default_rng(11 * d + K + n)
    means = 3.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    C0 = X[rng.choice(n, K, replace=False)].copy()
    if K >= 300:
        C0[5] = 100.0                                    # a centroid no sample is nearest to
    dt = _lib.Data(ctx, X)
    a = dt.kmeans_iterate(C0, max_steps, atol)
    la = dt.kmeans_labels()
    da = dt.kmeans_distances()
    dt.close()
    dt = _lib.Data(ctx, X)
    b = _kmeans_step_loop(dt, C0, max_steps, atol)
    lb = dt.kmeans_labels()
    db = dt.kmeans_distances()
    dt.close()
    assert a[0] == b[0] and a[1] == b[1]
    assert a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    assert np.array_equal(la, lb) and np.array_equal(da, db)
    if K >= 300:
        assert np.all(a[4][a[3] == 0] == 0.0)            # ML/KMeans.cpp:184
    if max_steps >= 2:
        # and the oracle's own fit from the same start: same number of steps, same decision, same labels
        km = oracle.KMeans(K)
        km.set_absolute_tolerance(atol)
        km.set_maximum_steps(max_steps)
        km.set_centroids_initialiser(oracle.FIXED, C0)
        assert km.fit(X) == a[1]
        assert km.steps_done == a[0]
        assert np.array_equal(km.labels, la)
        assert relerr(a[4], km.centroids) < 1e-12


@pytest.mark.parametrize("d,K,n", [(8, 16, 6000), (2, 7, 5000), (8, 256, 20000)])
def test_kmeans_iterate_entered_again_after_convergence(ctx, d, K, n):
    """The lagged loop launches a step ahead of the host's stopping test; when the test fires, that step must have left nothing
    behind: a second call from the converged centroids sees the same labels twice at once (2 steps), and labels, distances,
    centroids, counts and inertia are bit-identical before and after it."""
    from ml_amd import _lib
    rng = np.random.default_rng(5 * d + K)
    means = 3.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    C0 = X[rng.choice(n, K, replace=False)].copy()
    dt = _lib.Data(ctx, X)
    a = dt.kmeans_iterate(C0, 500, 0.0)
    assert a[1]                                           # converged on identical labels
    la, da = dt.kmeans_labels(), dt.kmeans_distances()
    b = dt.kmeans_iterate(a[4], 500, 0.0)
    assert b[0] == 2 and b[1]
    assert b[2] == a[2] and np.array_equal(b[3], a[3]) and np.array_equal(b[4], a[4])
    assert np.array_equal(dt.kmeans_labels(), la) and np.array_equal(dt.kmeans_distances(), da)
    # ... and a plain assignment under the same centroids changes no label
    inertia, changed = dt.kmeans_assign(a[4])
    assert changed == 0 and inertia == a[2]
    dt.close()
