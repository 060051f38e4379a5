"""The device-resident EM loop (em_resident.hip: the whole loop of EM::fit, ML/EM.cpp:143-170, in ONE launch for fits whose
iteration is a few microseconds -- the reference's own benchmark sizes, Benchmarks/bm_EM.cpp) against the three-launch loop of
runtime/em_loop.cpp (MLHIP_RESIDENT=0), which tests/test_gpu_iterate.py pins to the oracle: BIT-identical steps, convergence
flag, log-likelihood history, parameters and labels -- the kernel evaluates the same pass, the same sums in the same order and
the same closing arithmetic. Also against the oracle directly, a refinement hand-over, early convergence, max_steps = 1, and
the hand-off between workgroups under uneven load from a second stream."""
import threading

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
    mu0 = means + 0.2 * rng.standard_normal((K, d))
    return X, mu0


def _launches(ctx, name):
    return ctx.timing_get(name)[1]


def _same(a, b):
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[6], b[6])
    for x, y in zip(a[3:6], b[3:6]):
        assert np.array_equal(x, y)


SHAPES = [(2, 3, 10000), (2, 3, 100), (2, 3, 1000), (4, 3, 10000), (4, 4, 16384), (1, 16, 5000), (1, 1, 300), (2, 10, 16384),
          (3, 6, 7000), (6, 2, 3000), (3, 1, 64), (2, 5, 65)]


@pytest.mark.parametrize("d,K,n", SHAPES)
def test_resident_loop_is_bit_identical_to_the_three_launch_loop(ctx, d, K, n, monkeypatch):
    from ml_amd import _lib
    X, mu0 = _problem(d, K, n, 17 * d + K)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    pi0, S0 = np.full(K, 1.0 / K), np.stack([cov] * K)
    monkeypatch.setenv("MLHIP_RESIDENT", "0")
    ref_conv = dt.em_iterate(pi0, mu0, S0, 60, atol=1e-9)            # stops on the convergence test (or at 60)
    labels = dt.em_labels(K)
    resp = dt.em_responsibilities(K)
    ref_full = dt.em_iterate(pi0, mu0, S0, 7, 0.0, 0.0)              # tolerances 0: exactly max_steps iterations (ML/EM.cpp:163)
    ref_one = dt.em_iterate(pi0, mu0, S0, 1, 0.0, 0.0)
    monkeypatch.delenv("MLHIP_RESIDENT")
    ctx.timing_enable(True)
    ctx.timing_reset()
    got_conv = dt.em_iterate(pi0, mu0, S0, 60, atol=1e-9)
    assert _launches(ctx, "em_resident") == 1 and _launches(ctx, "em_fused") == 0      # the one launch, nothing else of the loop
    ctx.timing_enable(False)
    _same(got_conv, ref_conv)
    # the E-step state the loop leaves on the device is that of the LAST evaluated iteration, as after the launches
    assert np.array_equal(dt.em_labels(K), labels)
    assert np.array_equal(dt.em_responsibilities(K), resp)
    got_full = dt.em_iterate(pi0, mu0, S0, 7, 0.0, 0.0)
    assert got_full[0] == 7 and not got_full[1]
    _same(got_full, ref_full)
    _same(dt.em_iterate(pi0, mu0, S0, 1, 0.0, 0.0), ref_one)
    # a second fit on the same handle, other starting values (the ring and the exchange buffers are reused)
    mu1 = mu0 + 0.05
    again = dt.em_iterate(pi0, mu1, S0, 9, 0.0, 0.0)
    monkeypatch.setenv("MLHIP_RESIDENT", "0")
    _same(again, dt.em_iterate(pi0, mu1, S0, 9, 0.0, 0.0))
    dt.close()


@pytest.mark.parametrize("d,K,n", [(2, 3, 10000), (4, 3, 9000), (1, 8, 2000)])
def test_resident_loop_against_the_oracle(ctx, oracle, d, K, n):
    """The trajectory of ML/EM.cpp:143-170 spelt out with the oracle's expectation_step / maximisation_step."""
    from ml_amd import _lib
    X, mu0 = _problem(d, K, n, 5 * d + K)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    pi0, S0 = np.full(K, 1.0 / K), np.stack([cov] * K)
    steps, conv, ll, pi, mu, S, hist = dt.em_iterate(pi0, mu0, S0, 40, 1e-9, 1e-9)
    ref = oracle.EM(K)
    ref.set_parameters(mu0, S0, pi0)
    lls, old = [], None
    for step in range(40):
        ref.expectation_step(X)
        ref.maximisation_step(X)
        lls.append(ref.log_likelihood)
        if step > 0 and abs(lls[-1] - old) < 1e-9 + 1e-9 * max(abs(old), abs(lls[-1])):
            break
        old = lls[-1]
    assert steps == len(lls) and conv == (len(lls) < 40)
    assert np.max(np.abs(hist - np.array(lls)) / np.abs(np.array(lls))) < 1e-12        # log-likelihood: rel 1e-12
    rel = lambda a, b: np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))
    assert rel(pi, ref.mixing_probabilities) < 1e-11 and rel(mu, ref.means) < 1e-10 and rel(S, ref.covariances) < 1e-9
    ref.calculate_labels()
    assert np.array_equal(dt.em_labels(K), np.asarray(ref.labels))                     # labels: bit-exact
    dt.close()


@pytest.mark.parametrize("d", [2, 4])
def test_resident_loop_hands_a_flagged_iteration_to_the_host(ctx, d, monkeypatch):
    """A component far from the data mean and tight is flagged by the closing arithmetic: the kernel stops there, the host closes
    that iteration with its refinement pass (as the lagged loop does) and goes on -- same results as without the resident loop."""
    from ml_amd import _lib
    K, n = 3, 12000
    rng = np.random.default_rng(11)
    centres = np.array([[0.0] * d, [300.0] * d, [-200.0] * d])
    sig = np.array([1.0, 1e-3, 1e-2])
    comp = rng.integers(0, K, n)
    X = np.ascontiguousarray(centres[comp] + rng.standard_normal((n, d)) * sig[comp][:, None])
    mu0 = centres + 0.1 * sig[:, None] * rng.standard_normal((K, d))
    S0 = np.stack([np.eye(d) * v * 1.5 for v in sig ** 2])
    pi0 = np.full(K, 1.0 / K)
    dt = _lib.Data(ctx, X)
    ctx.timing_enable(True)
    ctx.timing_reset()
    got = dt.em_iterate(pi0, mu0, S0, 4, 0.0, 0.0)
    assert _launches(ctx, "em_resident") == 1
    assert _launches(ctx, "em_refine") >= 1                                              # the host's refinement pass ran
    ctx.timing_enable(False)
    monkeypatch.setenv("MLHIP_RESIDENT", "0")
    _same(got, dt.em_iterate(pi0, mu0, S0, 4, 0.0, 0.0))
    dt.close()


def test_hand_off_between_workgroups_under_uneven_load(ctx):
    """The per-iteration exchange (write-through stores, one arrival add per workgroup, sc1 loads behind the poll) while ANOTHER
    stream keeps a varying part of the chip busy: every fit must still give the bits of the quiet run. 200 iterations per fit."""
    from ml_amd import _lib
    d, K, n = 2, 3, 16000
    X, mu0 = _problem(d, K, n, 3)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    pi0, S0 = np.full(K, 1.0 / K), np.stack([cov] * K)
    ref = dt.em_iterate(pi0, mu0, S0, 200, 0.0, 0.0)
    other = _lib.Context()
    stop = threading.Event()

    def load():
        rng = np.random.default_rng(0)
        for size in (40000, 300000, 5000, 1200000) * 1000:
            if stop.is_set():
                break
            Y = rng.standard_normal((size, 8))
            od = _lib.Data(other, Y)
            od.kmeans_iterate(Y[:16].copy(), 3, 0.0)
            od.close()

    t = threading.Thread(target=load)
    t.start()
    try:
        for _ in range(25):
            _same(dt.em_iterate(pi0, mu0, S0, 200, 0.0, 0.0), ref)
    finally:
        stop.set()
        t.join()
    other.close()
    dt.close()
