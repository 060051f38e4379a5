#!/usr/bin/env python3
"""Ad-hoc: tests/test_gpu_random_sweep.py::test_random_shape on shapes drawn from another seed (argv[1])."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_random_sweep as T
from oracle import oracle_ctypes as oracle
from ml_amd import _lib
ctx = _lib.Context()
rng = np.random.default_rng(int(sys.argv[1]))
fails = 0
n_cases = 0
for d in list(rng.integers(1, 33, 40)) + list(rng.integers(33, 129, 12)) + list(rng.integers(129, 200, 3)):
    d = int(d)
    K = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 40, 47, 48, 49, 64, 65, 70, 96, 100]))
    if d > 64: K = min(K, 33)
    if d > 128: K = min(K, 5)
    n = int(rng.integers(max(8 * K, 70), 3000))
    seed = int(rng.integers(1 << 30))
    try:
        T.test_random_shape(ctx, oracle, d, K, n, seed)
    except AssertionError as e:
        fails += 1
        print("FAIL", d, K, n, seed, str(e)[:200])
    n_cases += 1
print("cases", n_cases, "failures", fails)
