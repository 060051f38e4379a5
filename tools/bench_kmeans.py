#!/usr/bin/env python3
"""Secondary benchmark: K-means steps/s at d=8, K=256 (BASELINE.json configs[4] shape; one GPU's share of N=100M is
12.5M samples). One step = mlhip_kmeans_step = assignment + exact update sums + reduce, data resident in HBM."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12_500_000)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--clusters", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from ml_amd import _lib, synth
    n, d, K = args.n, args.dim, args.clusters
    mix = synth.Mixture(d, K, seed=77, diagonal=True)
    X, _ = mix.sample(n)
    ctx = _lib.Context()
    data = _lib.Data(ctx, X)
    C = mix.means + 0.3 * np.random.default_rng(1).standard_normal((K, d))
    for _ in range(2):
        _, _, _, C = data.kmeans_step(C)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        inertia, changed, counts, C = data.kmeans_step(C)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        inertia, changed, counts, C = data.kmeans_step(C)
    ms, _ = ctx.timing_get("kmeans_assign")
    flops = float(n) * K * 3 * d          # N*K*3d per step
    print(json.dumps({"metric": "K-means steps/s", "N": n, "d": d, "K": K, "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt,
                      "kernel_ms": ms, "algorithmic_tflops": flops / (ms * 1e-3) / 1e12, "frac_of_78.6": flops / (ms * 1e-3) / 78.6e12,
                      "hbm_gbs_algorithmic": n * (8 * d + 4) / (ms * 1e-3) / 1e9, "inertia": inertia}))


if __name__ == "__main__":
    main()
