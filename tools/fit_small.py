import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd.cppyml import clustering as cl
from ml_amd import synth
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
