#!/usr/bin/env python3
"""Ad-hoc confidence run for the tiers above d = 64 (round 5): random shapes d in 65 .. 700, K in 1 .. 9 through mlhip_em_iterate with
the closing arithmetic on the device (em_close_big.hip) against the host closing -- one iteration from the same first records: the new
parameters bit for bit; three iterations: log-likelihood history 1e-13, parameters 1e-12 -- and one iteration against the ORACLE
(log-likelihood 1e-11, means 1e-10, covariances 1e-9).   usage: python tools/extra_big_sweep.py SEED [CASES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_close_big as T
from oracle import oracle_ctypes as oracle
from ml_amd import _lib

ctx = _lib.Context()
rng = np.random.default_rng(int(sys.argv[1]))
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fails = 0
for case in range(cases):
    d = int(rng.choice([rng.integers(65, 129), rng.integers(129, 300), rng.integers(300, 700), rng.choice([96, 128, 192, 256, 320, 512])]))
    K = int(rng.integers(1, max(2, min(10, 6000 // (d + 60) + 1))))       # every component gets more samples than dimensions: with fewer
    n = K * (d + 60) + int(rng.integers(0, 1000))                          # the covariance estimates are singular up to the 1e-15 ridge
                                                                           # (ML/EM.cpp:252) and the ulp of log() between the two closings is
                                                                           # amplified by a condition number of 1e15: nothing to compare
    X, mu0 = T._problem(d, K, n, int(rng.integers(1 << 30)))
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    S0, pi0 = np.stack([cov] * K), np.full(K, 1.0 / K)
    try:
        _, _, ll_h, pi_h, mu_h, S_h, _ = T._iterate(dt, pi0, mu0, S0, 1, False, device_records=False)
        _, _, ll_d, pi_d, mu_d, S_d, _ = T._iterate(dt, pi0, mu0, S0, 1, True, device_records=False)
        assert ll_d == ll_h and np.array_equal(pi_d, pi_h) and np.array_equal(mu_d, mu_h) and np.array_equal(S_d, S_h), "one iteration: bits"
        _, _, _, pi_h, mu_h, S_h, hist_h = T._iterate(dt, pi0, mu0, S0, 3, False)
        _, _, _, pi_d, mu_d, S_d, hist_d = T._iterate(dt, pi0, mu0, S0, 3, True)
        herr = np.max(np.abs(hist_d - hist_h) / np.abs(hist_h))
        assert herr < 1e-13, f"history {herr:.2e} (samples per component {n / K:.0f} at d = {d}; first-iteration ll equal: {hist_d[0] == hist_h[0]}, second: {abs(hist_d[1] - hist_h[1]) / abs(hist_h[1]):.1e})"
        sc = lambda a: max(1e-300, np.max(np.abs(a)))
        assert np.max(np.abs(mu_d - mu_h)) / sc(mu_h) < 1e-12 and np.max(np.abs(S_d - S_h)) / sc(S_h) < 1e-11, "three iterations"
        if d <= 400:
            em = oracle.EM(K)
            em.set_parameters(mu0, S0, pi0)
            em.expectation_step(X)
            em.maximisation_step(X)
            _, _, ll1, pi1, mu1, S1, _ = T._iterate(dt, pi0, mu0, S0, 1, True)
            assert abs(ll1 - em.log_likelihood) <= 1e-11 * abs(em.log_likelihood), "oracle ll"
            assert np.max(np.abs(mu1 - em.means)) / sc(em.means) < 1e-10 and np.max(np.abs(S1 - em.covariances)) / sc(em.covariances) < 1e-9, "oracle M-step"
    except AssertionError as e:
        fails += 1
        print("FAIL", d, K, n, str(e)[:200], flush=True)
    dt.close()
print("cases", cases, "failures", fails)
