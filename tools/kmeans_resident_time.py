#!/usr/bin/env python3
"""The K-means step loop on small blocks: one resident launch against the three launches per step (MLHIP_RESIDENT=0) -- wall time of
mlhip_kmeans_iterate and time per step, d = 2, K = 3 (Benchmarks/bm_KMeans.cpp's shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd import _lib

ctx = _lib.Context()
for n in (100, 1000, 2048, 4096):
    rng = np.random.default_rng(n)
    means = 3.0 * rng.standard_normal((3, 2))
    X = np.ascontiguousarray(means[rng.integers(0, 3, n)] + rng.standard_normal((n, 2)))
    c0 = X[rng.choice(n, 3, replace=False)].copy()
    dt = _lib.Data(ctx, X)
    row = []
    for mode in ("1", "0"):
        os.environ["MLHIP_RESIDENT"] = mode
        for _ in range(5): r = dt.kmeans_iterate(c0, 500, 0.0)
        t = []
        for _ in range(30):
            t0 = time.perf_counter(); r = dt.kmeans_iterate(c0, 500, 0.0); t.append(time.perf_counter() - t0)
        row.append((np.median(t) * 1e6, r[0]))
    del os.environ["MLHIP_RESIDENT"]
    print("N=%6d d=2 K=3: resident %.1f us (%d steps, %.2f us/step) | three launches %.1f us (%.2f us/step)" %
          (n, row[0][0], row[0][1], row[0][0] / row[0][1], row[1][0], row[1][0] / row[1][1]), flush=True)
    dt.close()
