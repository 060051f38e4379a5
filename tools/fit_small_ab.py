"""Whole-fit wall time of the reference's bm_EM.cpp-sized case (N=10k, d=4, K=3, KPP, 50 iterations) through ml_amd.cppyml: median of
40 fits. MLHIP_LIBRARY selects another build of the library for A/B runs (e.g. a round-3 build); MLHIP_ONE_LAUNCH=0/1 and MLHIP_GRAPH=0/1 the launch form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ml_amd.cppyml import clustering as cl
from ml_amd import synth
mix = synth.Mixture(4, 3, seed=3)
X, _ = mix.sample(10000)


def fit():
    em = cl.EM(3); em.set_seed(1); em.set_means_initialiser(cl.KPP()); em.set_maximum_steps(50)
    em.set_absolute_tolerance(0.0); em.set_relative_tolerance(0.0)
    t0 = time.perf_counter(); em.fit(X); return time.perf_counter() - t0


for _ in range(3):
    fit()
ts = sorted(fit() for _ in range(40))
print("%s ONE_LAUNCH=%s GRAPH=%s: median %.3f ms, min %.3f ms" % (os.environ.get("MLHIP_LIBRARY", "current")[-20:], os.environ.get("MLHIP_ONE_LAUNCH", "-"), os.environ.get("MLHIP_GRAPH", "-"), ts[20] * 1e3, ts[0] * 1e3))
