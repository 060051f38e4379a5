#!/usr/bin/env python3
"""Ad-hoc: K-means steps on random shapes around the few-cluster rule (K <= 24, d <= 32, N just above 2^21): the direct-form kernel
with its accumulator copies (default there) against the matrix-core search (MLHIP_KMEANS=mfma) -- labels, distances, counts and new
centroids must be the same bits -- and against the oracle's assignment on a prefix.   usage: tools/extra_kmeans_sweep.py SEED [CASES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_ctypes as oracle
from ml_amd import _lib

ctx = _lib.Context()
rng = np.random.default_rng(int(sys.argv[1]))
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fails = 0
for c in range(cases):
    d = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32]))
    K = int(rng.integers(1, 26))
    n = (1 << 21) + int(rng.integers(0, 5000))
    C = 3.0 * rng.standard_normal((K, d))
    X = np.ascontiguousarray(C[rng.integers(0, K, n)] + rng.standard_normal((n, d)))
    C0 = C + 0.5 * rng.standard_normal((K, d))
    dt = _lib.Data(ctx, X)
    os.environ.pop("MLHIP_KMEANS", None)
    a = dt.kmeans_step(C0); la, da = dt.kmeans_labels(), dt.kmeans_distances()
    os.environ["MLHIP_KMEANS"] = "mfma"
    b = dt.kmeans_step(C0); lb, db = dt.kmeans_labels(), dt.kmeans_distances()
    os.environ.pop("MLHIP_KMEANS", None)
    bad = []
    if not (np.array_equal(la, lb) and np.array_equal(da, db)): bad.append("labels/distances differ between the kernels")
    if not (np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])): bad.append("counts/centroids differ")   # (a[1]: labels changed since the previous call)
    if not abs(a[0] - b[0]) <= 1e-13 * abs(b[0]): bad.append("inertia")
    m = 20000
    km = oracle.KMeans(K)
    km.set_centroids(C0, m)
    km.assignment_step(X[:m])
    if not np.array_equal(la[:m], km.labels): bad.append("labels differ from the oracle")
    if bad:
        fails += 1
        print("FAIL d=%d K=%d n=%d: %s" % (d, K, n, "; ".join(bad)))
    dt.close()
print("cases", cases, "failures", fails)
