"""Where the wall time of a bm_EM-sized fit (N=10k, d=4, K=3, 50 iterations) goes: data handle (upload, transpose, shift), first
(cold) and second (warm) mlhip_em_iterate on the handle, labels, release -- medians over 30 handles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd import _lib, synth

n, d, K = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (10000, 4, 3)))
mix = synth.Mixture(d, K, seed=3)
X, _ = mix.sample(n)
ctx = _lib.Context()
rows = []
for rep in range(33):
    t = [time.perf_counter()]
    dt = _lib.Data(ctx, X); t.append(time.perf_counter())
    _, cov = dt.sample_covariance(); t.append(time.perf_counter())
    pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
    dt.em_iterate(pi, mu, S, 50); t.append(time.perf_counter())
    dt.em_iterate(pi, mu, S, 50); t.append(time.perf_counter())
    dt.em_labels(K); t.append(time.perf_counter())
    dt.close(); t.append(time.perf_counter())
    if rep >= 3:
        rows.append(np.diff(t))
m = np.median(np.array(rows), axis=0) * 1e6
print("N=%d d=%d K=%d  us: handle %.0f | sample covariance %.0f | em_iterate cold %.0f | warm %.0f | labels %.0f | release %.0f" % (n, d, K, *m))
