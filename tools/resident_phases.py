#!/usr/bin/env python3
"""Per-phase times of the device-resident EM loop (em_resident.hip; MLHIP_RESIDENT_PROFILE=1: workgroup 0 stamps the 100 MHz clock
at the phase boundaries of every iteration, the runtime prints the averages on stderr) next to the wall time per iteration of the
resident loop and of the three-launch loop (MLHIP_RESIDENT=0), at the reference's own benchmark shapes (Benchmarks/bm_EM.cpp)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd import _lib, synth

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(100, 2, 3), (1000, 2, 3), (10000, 2, 3), (10000, 4, 3), (65536, 2, 3), (16384, 1, 21)]
ctx = _lib.Context()
for n, d, K in shapes:
    mix = synth.Mixture(d, K, seed=3)
    X, _ = mix.sample(n)
    dt = _lib.Data(ctx, X)
    _, cov = dt.sample_covariance()
    pi, mu, S = np.full(K, 1.0 / K), mix.initial_means(), np.stack([cov] * K)
    out = {}
    for label, env in (("resident", None), ("three launches", "0")):
        if env is None:
            os.environ.pop("MLHIP_RESIDENT", None)
        else:
            os.environ["MLHIP_RESIDENT"] = env
        dt.em_iterate(pi, mu, S, 5)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            dt.em_iterate(pi, mu, S, 400)
            best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
        out[label] = best
    os.environ.pop("MLHIP_RESIDENT", None)
    print(f"N={n} d={d} K={K}: " + ", ".join(f"{k} {v:.2f} us/iteration" for k, v in out.items()), flush=True)
    if os.environ.get("MLHIP_RESIDENT_PROFILE") == "1":
        dt.em_iterate(pi, mu, S, 400)            # (its stderr line: the phases)
    dt.close()
