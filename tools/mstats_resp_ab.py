#!/usr/bin/env python3
"""Statistics kernel on plain responsibilities against the same kernel normalising log-responsibilities while staging:
what the exponentials of the staging phase cost at a shape (upper bound for a pre-normalised form).   python tools/mstats_resp_ab.py N d K"""
import sys, numpy as np
sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
from ml_amd import _lib as L

N, d, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1000000, 128, 32)
rng = np.random.default_rng(5)
centres = rng.normal(scale=3.0, size=(K, d))
lab = rng.integers(0, K, size=N)
X = np.ascontiguousarray(centres[lab] + rng.normal(size=(N, d)))
ctx = L.Context()
data = L.Data(ctx, X)
mix = np.full(K, 1.0 / K); means = centres.copy(); covs = np.stack([np.eye(d)] * K)
ctx.timing_enable(True)
for rep in range(3):
    ctx.timing_reset()
    for _ in range(4): data.em_step(mix, means, covs)
    ms, cnt = ctx.timing_get("em_mstats")
    print("E-step's log-responsibilities: em_mstats %.3f ms (%d launches)" % (ms / max(cnt, 1), cnt), flush=True)
R = data.em_responsibilities(K)
for rep in range(3):
    ctx.timing_reset()
    for _ in range(4): data.em_maximisation_from(R)
    ms, cnt = ctx.timing_get("em_mstats")
    print("plain responsibilities:        em_mstats %.3f ms (%d launches)" % (ms / max(cnt, 1), cnt), flush=True)
