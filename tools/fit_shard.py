#!/usr/bin/env python3
"""Wall time of whole fits at one GPU's share of the BASELINE configurations, through ml_amd.cppyml.clustering from a host array:
KMeans.fit (Forgy, the reference's default initialiser) at N=12.5M, d=8, K=256 and EM.fit (Forgy + ClosestCentroid) at N=1.25M, d=32, K=64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_amd.cppyml import clustering as cl
from ml_amd import synth

X, _ = synth.Mixture(8, 256, seed=77, diagonal=True).sample(12_500_000)
for rep in range(2):
    km = cl.KMeans(256)
    km.set_seed(1)
    km.set_maximum_steps(50)
    t0 = time.perf_counter()
    km.fit(X)
    t = time.perf_counter() - t0
print("KMeans.fit N=12.5M d=8 K=256, Forgy, %d steps: %.1f ms" % (km.steps_done, t * 1e3))
del X
X, _ = synth.Mixture(32, 64, seed=7).sample(1_250_000)
for rep in range(2):
    em = cl.EM(64)
    em.set_seed(1)
    em.set_maximum_steps(50)
    em.set_absolute_tolerance(0.0)
    em.set_relative_tolerance(0.0)
    t0 = time.perf_counter()
    em.fit(X)
    t = time.perf_counter() - t0
print("EM.fit N=1.25M d=32 K=64, Forgy + ClosestCentroid, %d iterations: %.1f ms" % (em.steps_done, t * 1e3))
