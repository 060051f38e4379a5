import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd.cppyml import clustering as cl
from ml_amd import synth
"""Whole `EM.fit` / `KMeans.fit` calls through cppyml (upload + initialisation + 50 iterations + labels): the FIRST call of a shape in this
process against the later ones. The library's process-wide context is created by a throw-away fit of four 1-d points first and timed on
its own line: it is paid once per process whatever is fitted (device initialisation), and round 4's "first fit 209 ms" mixed it into
the first shape's number."""
t0 = time.perf_counter()
_km = cl.KMeans(1)
_km.fit(np.array([[0.0], [1.0], [2.0], [3.0]]))
print(f"context creation (+ a 4-point fit): {(time.perf_counter() - t0) * 1e3:.1f} ms")
for n, d, K in ((10_000, 4, 3), (100_000, 16, 8), (1_000_000, 16, 16)):
    mix = synth.Mixture(d, K, seed=3)
    X, _ = mix.sample(n)
    ts = []
    for rep in range(6):
        em = cl.EM(K)
        em.set_seed(42)
        em.set_maximum_steps(50)
        em.set_absolute_tolerance(0.0)
        em.set_relative_tolerance(0.0)
        t0 = time.perf_counter()
        em.fit(X)
        ts.append(time.perf_counter() - t0)
    tk = []
    for rep in range(6):
        km = cl.KMeans(K)
        km.set_seed(42)
        km.set_maximum_steps(50)
        t0 = time.perf_counter()
        km.fit(X)
        tk.append(time.perf_counter() - t0)
    print(f"N={n} d={d} K={K}: EM.fit 50 iterations: first {ts[0]*1e3:.1f} ms, then {np.median(ts[1:])*1e3:.2f} ms; KMeans.fit: first {tk[0]*1e3:.1f} ms, then {np.median(tk[1:])*1e3:.2f} ms")
