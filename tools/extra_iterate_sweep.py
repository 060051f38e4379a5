#!/usr/bin/env python3
"""Ad-hoc: the multi-iteration parity check of tests/test_gpu_mstats_plan.py on random (d, K) drawn from another seed (argv[1])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_mstats_plan as T
from oracle import oracle_ctypes as oracle

rng = np.random.default_rng(int(sys.argv[1]))
fails = cases = 0
for _ in range(24):
    d = int(rng.choice([8, 10, 12, 13, 16, 17, 20, 24, 28, 32, 36, 40, 48]))
    K = int(rng.choice([2, 7, 16, 17, 24, 32, 33, 40, 48, 50, 64]))
    if d > 32: K = min(K, 24)
    try:
        T.test_row_block_groups_and_balanced_units_match_the_oracle(oracle, d, K)
    except AssertionError as e:
        fails += 1
        print("FAIL", d, K, str(e)[:300])
    cases += 1
print("cases", cases, "failures", fails)
