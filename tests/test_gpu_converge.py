"""Converge runs at the headline dimension and component count (d = 32, K = 64) against the oracle -- north_star asks for
bit-exact assignments AT CONVERGENCE; SURVEY 8(d) for a separate tolerance-1e-10 run (ML/EM.cpp:161-168: the convergence test;
:289-304: calculate_labels runs only then).

1. n = 24 000: the oracle's own loop (expectation_step / maximisation_step + the reference's test) gives the log-likelihood
   TRAJECTORY, the step count and the labels; the facade's EM.fit and mlhip_em_iterate (the path the bench runs: FOLD E-step,
   self-normalising statistics kernel, device closing) must reproduce them: same number of steps, every log-likelihood to 1e-12,
   means / covariances to 1e-10, labels bit-exact.
2. N = 1 000 000: the fit with the E-step's FOLD form on / off and the self-normalising statistics kernel on / off (one process
   per setting: the switches are read once) ends in the same number of steps with IDENTICAL labels, and the oracle's E-step on the
   first 50 000 rows, at the converged parameters, labels those rows the same way."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, K = 32, 64


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


def test_converge_run_at_d32_K64_matches_the_oracle_trajectory(oracle):
    from ml_amd import _lib, synth
    from ml_amd.cppyml import clustering as cl
    n, atol, rtol, max_steps = 24000, 1e-10, 1e-10, 200
    mix = synth.Mixture(D, K, seed=31)
    X, _ = mix.sample(n)
    start = mix.initial_means()

    # the oracle: EM::fit's loop spelt out (ML/EM.cpp:127-170), so that the trajectory is visible
    cov = np.ascontiguousarray(oracle.sample_covariance(X))
    ref = oracle.EM(K)
    ref.set_parameters(start, np.stack([cov] * K), np.full(K, 1.0 / K))
    lls, old, conv = [], None, False
    for step in range(max_steps):
        ref.expectation_step(X)
        ref.maximisation_step(X)
        ll = ref.log_likelihood
        lls.append(ll)
        if step > 0 and abs(ll - old) < atol + rtol * max(abs(old), abs(ll)):
            ref.calculate_labels()
            conv = True
            break
        old = ll
    assert conv and len(lls) >= 4
    labels_ref = np.asarray(ref.labels)

    # mlhip_em_iterate from the same start
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    _, cov_dev = dt.sample_covariance()
    assert relerr(cov_dev, cov) < 1e-12
    steps, conv_b, ll_b, pi_b, mu_b, S_b, hist = dt.em_iterate(np.full(K, 1.0 / K), start, np.stack([cov_dev] * K), max_steps, atol, rtol)
    assert conv_b and steps == len(lls)
    assert np.max(np.abs(hist - np.array(lls)) / np.abs(np.array(lls))) <= 1e-12
    assert relerr(pi_b, ref.mixing_probabilities) <= 1e-11 and relerr(mu_b, ref.means) <= 1e-10
    for k in range(K):
        assert relerr(S_b[k], ref.covariances[k]) <= 1e-10, k
    assert np.array_equal(dt.em_labels(K), labels_ref)
    assert np.max(np.abs(dt.em_responsibilities(K) - ref.responsibilities)) <= 1e-11
    dt.close()
    ctx.close()

    # the drop-in class
    em = cl.EM(K)
    em.set_means_initialiser(cl.FixedCentroids(start))
    em.set_absolute_tolerance(atol)
    em.set_relative_tolerance(rtol)
    em.set_maximum_steps(max_steps)
    assert em.fit(X)
    assert em.steps_done == len(lls)
    assert abs(em.log_likelihood - lls[-1]) <= 1e-12 * abs(lls[-1])
    assert np.array_equal(np.asarray(em.labels), labels_ref)
    assert relerr(em.means.T, ref.means) <= 1e-10


CHILD = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from ml_amd import _lib, synth
D, K, n = 32, 64, 1000000
mix = synth.Mixture(D, K, seed=32)
X, _ = mix.sample(n)
ctx = _lib.Context()
dt = _lib.Data(ctx, X)
_, cov = dt.sample_covariance()
steps, conv, ll, pi, mu, S, hist = dt.em_iterate(np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K), 200, 1e-10, 1e-10)
labels = dt.em_labels(K)
# labels of ALL rows at the converged parameters (one more E-step), for the oracle comparison on a prefix
dt.em_expectation(pi, mu, S)
at_final = dt.em_labels(K)
np.savez(sys.argv[1], pi=pi, mu=mu, S=S, prefix=at_final[:50000])
print("CHILD " + json.dumps({"steps": steps, "conv": conv, "ll": float(ll).hex(), "plan": dt.em_plan(K),
                             "labels": hashlib.sha256(labels.tobytes()).hexdigest(),
                             "at_final": hashlib.sha256(at_final.tobytes()).hexdigest()}))
"""


def test_full_size_converge_run_is_independent_of_the_kernel_variants(oracle, tmp_path):
    from ml_amd import synth
    results = {}
    for name, env in (("default", {}), ("no_fold", {"MLHIP_ESTEP_FOLD": "0"}), ("no_self_norm", {"MLHIP_SELF_NORM": "0"}),
                      ("neither", {"MLHIP_ESTEP_FOLD": "0", "MLHIP_SELF_NORM": "0"})):
        out = os.path.join(tmp_path, name + ".npz")
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, out], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("CHILD ")][0]
        results[name] = (json.loads(line[6:]), out)
    base = results["default"][0]
    assert base["conv"] and base["plan"]["self_norm"] and base["plan"]["matrix_estep"]
    assert not results["no_self_norm"][0]["plan"]["self_norm"]
    for name, (r, _) in results.items():
        assert r["conv"] and r["steps"] == base["steps"], name
        assert r["labels"] == base["labels"] and r["at_final"] == base["at_final"], name       # bit-exact labels, 1M rows
        assert abs(float.fromhex(r["ll"]) - float.fromhex(base["ll"])) <= 1e-12 * abs(float.fromhex(base["ll"])), name
    # the oracle on the first 50 000 rows of the same data, at the converged parameters
    z = np.load(results["default"][1])
    X, _ = synth.Mixture(D, K, seed=32).sample(1000000)
    ref = oracle.EM(K)
    ref.set_parameters(z["mu"], z["S"], z["pi"])
    ref.expectation_step(np.ascontiguousarray(X[:50000]))
    ref.calculate_labels()
    assert np.array_equal(np.asarray(ref.labels), z["prefix"])
