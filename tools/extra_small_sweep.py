#!/usr/bin/env python3
"""Ad-hoc: random small-dimension EM shapes (d = 1..8, K = 1..40, ragged N) through every form of the fused E+M kernel -- default,
vector-unit form wherever it is built (MLHIP_FUSED_VALU=2), matrix-core form with the LDS feed (MLHIP_FUSED_VALU=0
MLHIP_FUSED_SFEED=0) and with the scalar feed (=1), and the two-kernel path (MLHIP_FUSED=0) -- each against the oracle's E- and
M-step.   usage: tools/extra_small_sweep.py SEED [CASES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_ctypes as oracle
from ml_amd import _lib

FORMS = {"default": {}, "valu": {"MLHIP_FUSED_VALU": "2"}, "mc-lds": {"MLHIP_FUSED_VALU": "0", "MLHIP_FUSED_SFEED": "0"},
         "mc-scalar": {"MLHIP_FUSED_VALU": "0", "MLHIP_FUSED_SFEED": "1"}, "two-kernel": {"MLHIP_FUSED": "0"}}
KEYS = ("MLHIP_FUSED_VALU", "MLHIP_FUSED_SFEED", "MLHIP_FUSED")


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


ctx = _lib.Context()
rng = np.random.default_rng(int(sys.argv[1]))
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
fails = 0
for c in range(cases):
    d = int(rng.integers(1, 9))
    K = int(rng.integers(1, 41))
    n = int(rng.integers(max(60 * K, 70), 6000)) if c % 10 else int(rng.integers(1 << 19, (1 << 20) + 5000))
    means = 3.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.standard_normal((n, d)) * rng.uniform(0.5, 1.5) + rng.uniform(-30, 30))
    mu0 = means + X.mean(axis=0) - means.mean(axis=0) + 0.3 * rng.standard_normal((K, d))
    A = rng.standard_normal((K, d, d)) * 0.3
    S0 = np.stack([np.eye(d) * rng.uniform(0.6, 1.8) + A[k] @ A[k].T for k in range(K)])
    pi0 = rng.uniform(0.3, 1.7, K); pi0 /= pi0.sum()
    em = oracle.EM(K)
    em.set_parameters(mu0, S0, pi0)
    em.expectation_step(X)
    ll, R = em.log_likelihood, em.responsibilities.copy()
    em.maximisation_step(X)
    ref = (em.mixing_probabilities, em.means, em.covariances)
    dt = _lib.Data(ctx, X)
    for name, env in FORMS.items():
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        got = dt.em_step(pi0, mu0, S0)
        Rg = dt.em_responsibilities(K)
        bad = []
        if not abs(got[0] - ll) <= 1e-12 * abs(ll): bad.append("ll %.3e" % abs(got[0] / ll - 1))
        if not np.max(np.abs(Rg - R)) < 1e-12: bad.append("resp %.3e" % np.max(np.abs(Rg - R)))
        for a, b, tol, what in zip(got[1:], ref, (1e-11, 1e-11, 1e-9), ("mixing", "means", "covs")):
            if not relerr(a, b) < tol: bad.append("%s %.3e" % (what, relerr(a, b)))
        if bad:
            fails += 1
            print("FAIL d=%d K=%d n=%d form=%s: %s" % (d, K, n, name, ", ".join(bad)))
    dt.close()
for k in KEYS:
    os.environ.pop(k, None)
print("cases", cases, "x", len(FORMS), "forms; failures", fails)
