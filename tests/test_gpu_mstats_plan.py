"""The statistics kernel's decompositions added in round 3 (ml_amd/csrc/device/em_mstats_wide.hip): row-block groups of THREE
16-component blocks (K = 33..48 in one group, K = 65..96 in two, ...) and the balanced dealing of (column block, row block) units
at d = 12 / 16. One E + M iteration through mlhip_em_step AND three through mlhip_em_iterate (self-normalising form where one group
holds all components) against the oracle (ML/EM.cpp:172-287), plus the balanced form against the plain one."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


def _problem(d, K, n, seed):
    rng = np.random.default_rng(seed)
    means = 2.5 * rng.standard_normal((K, d)) + rng.uniform(-3, 3, d)
    X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.uniform(0.6, 1.4, d) * rng.standard_normal((n, d)))
    mu0 = means + 0.3 * rng.standard_normal((K, d))
    S0 = np.stack([np.cov(X.T) * rng.uniform(0.2, 0.5) + 0.2 * np.eye(d) for _ in range(K)])
    pi0 = rng.dirichlet(np.ones(K) * 6)
    return X, pi0, mu0, S0


@pytest.mark.parametrize("d,K", [(12, 33), (12, 48), (12, 64), (16, 17), (16, 32), (16, 40), (16, 48), (16, 64), (16, 80), (24, 48),
                                 (32, 48), (32, 96), (40, 48), (48, 40), (72, 33), (16, 130)])
def test_row_block_groups_and_balanced_units_match_the_oracle(oracle, d, K):
    from ml_amd import _lib
    n = K * max(40, 4 * d) + 777          # enough samples per component for well-conditioned covariances
    X, pi0, mu0, S0 = _problem(d, K, n, 1000 * d + K)
    ctx = _lib.Context()
    dt = _lib.Data(ctx, X)
    ll, pi1, mu1, S1 = dt.em_step(pi0, mu0, S0)
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    assert abs(ll - em.log_likelihood) <= 1e-12 * abs(em.log_likelihood)
    em.maximisation_step(X)
    assert relerr(pi1, em.mixing_probabilities) < 1e-11 and relerr(mu1, em.means) < 1e-11 and relerr(S1, em.covariances) < 1e-9

    # the fit loop (device closing; self-normalising statistics kernel when the plan says so): three iterations
    steps, conv, ll3, pi3, mu3, S3, hist = dt.em_iterate(pi0, mu0, S0, 3, 0.0, 0.0)
    plan = dt.em_plan(K)
    assert plan["self_norm"] == (K <= 64 and d >= 10)       # matrix-core E-step (padded d >= 12) + wide statistics kernel, one row-block group
    lls = [em.log_likelihood]
    for _ in range(2):
        em.expectation_step(X)
        em.maximisation_step(X)
        lls.append(em.log_likelihood)
    assert steps == 3 and np.max(np.abs(hist - np.array(lls)) / np.abs(np.array(lls))) <= 1e-12
    assert relerr(pi3, em.mixing_probabilities) < 1e-10 and relerr(mu3, em.means) < 1e-10 and relerr(S3, em.covariances) < 1e-8
    em.expectation_step(X)
    em.calculate_labels()
    dt.em_expectation(pi3, mu3, S3)
    mism = np.flatnonzero(dt.em_labels(K) != np.asarray(em.labels))
    R = em.responsibilities
    for i in mism:          # parameters differ in the last digits after three iterations: only near-ties may flip
        top = np.sort(R[i])[-2:]
        assert top[1] - top[0] < 1e-8, i
    dt.close()
    ctx.close()


CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from test_gpu_mstats_plan import _problem
from ml_amd import _lib
d, K = int(sys.argv[1]), int(sys.argv[2])
X, pi0, mu0, S0 = _problem(d, K, 30011, 5)
ctx = _lib.Context()
dt = _lib.Data(ctx, X)
out = dt.em_iterate(pi0, mu0, S0, 2, 0.0, 0.0)
np.savez(sys.argv[3], ll=out[2], pi=out[3], mu=out[4], S=out[5])
"""


@pytest.mark.parametrize("d,K", [(16, 64), (16, 48), (16, 20), (12, 64), (12, 40)])
def test_balanced_and_plain_dealing_agree(tmp_path, d, K):
    """Both forms add the same products in the same order into every accumulator tile -- which wave holds a tile does not enter
    the arithmetic: results are bit-identical."""
    got = {}
    for b in ("0", "1"):
        out = os.path.join(tmp_path, b + ".npz")
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(d), str(K), out], env=dict(os.environ, MLHIP_MSTATS_BALANCED=b),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        got[b] = np.load(out)
    for key in ("ll", "pi", "mu", "S"):
        assert np.array_equal(got["0"][key], got["1"][key]), key
