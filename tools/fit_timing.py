#!/usr/bin/env python3
"""End-to-end timing of the drop-in API at the headline size: ml_amd.cppyml.clustering.EM(...).fit(X) from a host numpy array
(upload over PCIe + initialisation + iterations + labels), next to the per-iteration figure bench.py reports."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ml_amd import synth
    from ml_amd.cppyml import clustering
    n, d, K = 10_000_000, 32, 64
    mix = synth.Mixture(d, K)
    X, _ = mix.sample(n)
    out = {}
    for name, make_init in (("fixed", lambda: clustering.FixedCentroids(mix.initial_means())), ("kpp", clustering.KPP),
                            ("forgy", clustering.Forgy)):
        em = clustering.EM(K)
        em.set_means_initialiser(make_init())
        em.set_absolute_tolerance(1e-10)
        em.set_relative_tolerance(1e-10)
        em.set_maximum_steps(30)
        em.set_seed(1)
        t0 = time.perf_counter()
        conv = em.fit(X)
        t1 = time.perf_counter()
        out[name] = {"seconds": t1 - t0, "converged": conv, "steps": em.steps_done, "log_likelihood": em.log_likelihood}
        t0 = time.perf_counter()
        labels = em.labels
        out[name]["labels_seconds"] = time.perf_counter() - t0
        if name == "fixed":
            t0 = time.perf_counter()
            resp = em.responsibilities
            out[name]["responsibilities_seconds"] = time.perf_counter() - t0
            out[name]["responsibilities_gb"] = resp.nbytes / 1e9
            del resp
        del em
    km = clustering.KMeans(256)
    X8 = np.ascontiguousarray(X[:, :8])
    km.set_centroids_initialiser(clustering.Forgy())
    km.set_maximum_steps(20)
    t0 = time.perf_counter()
    conv = km.fit(X8)
    out["kmeans_forgy_d8_K256"] = {"seconds": time.perf_counter() - t0, "converged": conv, "steps": km.steps_done}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
