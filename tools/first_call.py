"""Cold-start costs of a fresh process: import, context, first data handle, first EM iteration loop of a shape, the same again,
the first loop of ANOTHER shape (new kernels: their code objects load on first use)."""
import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd import _lib, synth
t = [time.perf_counter()]
ctx = _lib.Context(); t.append(time.perf_counter())


def fit(n, d, K):
    mix = synth.Mixture(d, K, seed=3)
    X, _ = mix.sample(n)
    a = time.perf_counter()
    dt = _lib.Data(ctx, X); b = time.perf_counter()
    _, cov = dt.sample_covariance(); c = time.perf_counter()
    dt.em_iterate(np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K), 20); e = time.perf_counter()
    dt.kmeans_iterate(mix.initial_means(), 20, 0.0); f = time.perf_counter()
    dt.close()
    return (b - a) * 1e3, (c - b) * 1e3, (e - c) * 1e3, (f - e) * 1e3


print("import %.0f ms, context %.0f ms" % ((t[0] - t0) * 1e3, (t[1] - t[0]) * 1e3))
for shape in ((10000, 4, 3), (10000, 4, 3), (20000, 16, 8), (20000, 16, 8), (20000, 32, 70), (20000, 2, 300)):
    print("N=%d d=%d K=%d: handle %.1f ms, covariance %.1f ms, em_iterate(20) %.1f ms, kmeans_iterate %.1f ms" % (*shape, *fit(*shape)))
