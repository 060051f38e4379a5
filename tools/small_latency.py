#!/usr/bin/env python3
"""Per-iteration latency at the reference's own CPU-sized configuration (bm_EM.cpp: N=10k, d=4, K=3) and a mid-size one."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_amd import _lib, synth

out = {}
ctx = _lib.Context()
for n, d, K in ((10_000, 4, 3), (100_000, 8, 8), (100_000, 16, 8), (20_000, 32, 4), (1_000_000, 16, 16)):
    mix = synth.Mixture(d, K, seed=3)
    X, _ = mix.sample(n)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
    for _ in range(5):
        ll, pi, mu, S = dt.em_step(pi, mu, S)
    t0 = time.perf_counter()
    for _ in range(200):
        ll, pi, mu, S = dt.em_step(pi, mu, S)
    out[f"N={n},d={d},K={K}"] = {"us_per_iteration": (time.perf_counter() - t0) / 200 * 1e6}
    dt.em_iterate(pi, mu, S, 5)          # first call: ring / pinned buffers, events, the closing kernel's code object (0.5 - 15 ms)
    t0 = time.perf_counter()
    dt.em_iterate(pi, mu, S, 200)
    out[f"N={n},d={d},K={K}"]["us_per_iteration_em_iterate"] = (time.perf_counter() - t0) / 200 * 1e6
    Cc = mu.copy()
    for _ in range(5):
        _, _, _, Cc = dt.kmeans_step(Cc)
    t0 = time.perf_counter()
    for _ in range(200):
        _, _, _, Cc = dt.kmeans_step(Cc)
    out[f"N={n},d={d},K={K}"]["us_per_kmeans_step"] = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    done = 0
    while done < 200:
        done += dt.kmeans_iterate(Cc, 200 - done)[0]
    out[f"N={n},d={d},K={K}"]["us_per_kmeans_step_iterate"] = (time.perf_counter() - t0) / 200 * 1e6
    dt.close()
print(json.dumps(out))
