#!/usr/bin/env python3
"""Generates the golden fixtures in this directory. Run once, in the build container:

    python tests/golden/make_golden.py

Needs scikit-learn + scipy (third-party pins; the reference itself needs Eigen and cannot be built or
imported here, see oracle/README.md). Fixtures are pure data (inputs + expected outputs); they travel to
the GPU box, this script's dependencies need not.

Fixtures
--------
mousie_sklearn.npz      the data set and the scikit-learn pin of the reference's own Python test
                        (cppyml/tests/test_clustering.py:15-67): X from np.random.seed(999), and
                        GaussianMixture(3, tol=1e-10, max_iter=1000, random_state=999, n_init=1,
                        reg_covar=1e-15).fit(X).score(X).
em_onestep_*.npz        one E+M iteration from explicit parameters (pi0, mu0, Sigma0): mean log-likelihood
                        of the E-step, responsibilities/labels under the initial parameters (scipy, independent
                        of sklearn), and sklearn's M-step outputs (weights_, means_, covariances_ with
                        reg_covar=1e-15 == the reference's ridge, ML/EM.cpp:252-256).
em_onestep_diag_*.npz   the same for DIAGONAL covariances (BASELINE.json configs[1]; an extension -- the reference has no
                        diagonal mode, ML/EM.hpp:175 -- so this third-party pin is the only one): scipy E-step under
                        (pi0, mu0, var0) and sklearn covariance_type='diag' M-step outputs (variances K x d).
kmeans_onestep_*.npz    one Lloyd step from explicit centroids: labels/inertia (numpy, direct squared
                        distances as in ML/KMeans.cpp:153-165) and sklearn's updated centres.
"""
import os
import warnings

import numpy as np
import scipy.special
import scipy.stats
import sklearn.cluster
import sklearn.mixture

HERE = os.path.dirname(os.path.abspath(__file__))


def mousie_numpy(seed=999, sample_size=1000):
    """The 'mousie' set of the reference's Python test: same legacy-RandomState draw order."""
    np.random.seed(seed)
    face_radius, ear_radius, ear_angle = 1.0, 0.3, np.pi / 4
    radii = (face_radius, ear_radius, ear_radius)
    ear_weight = 2
    weights = np.array([face_radius ** 2, ear_weight * ear_radius ** 2, ear_weight * ear_radius ** 2])
    probabilities = weights / np.sum(weights)
    indices = np.random.choice(np.arange(3), sample_size, p=probabilities)
    cx = [0, (face_radius + ear_radius) * np.sin(-ear_angle), (face_radius + ear_radius) * np.sin(ear_angle)]
    cy = [0, (face_radius + ear_radius) * np.cos(-ear_angle), (face_radius + ear_radius) * np.cos(ear_angle)]
    data = np.empty((sample_size, 2))
    for i in range(sample_size):
        k = indices[i]
        phi = np.random.rand() * 2 * np.pi
        r = np.sqrt(np.random.rand()) * radii[k]
        data[i, 0] = cx[k] + r * np.cos(phi)
        data[i, 1] = cy[k] + r * np.sin(phi)
    return data


def make_mousie():
    X = mousie_numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        gmm = sklearn.mixture.GaussianMixture(3, tol=1e-10, max_iter=1000, random_state=999, n_init=1, reg_covar=1e-15)
        gmm.fit(X)
    np.savez(os.path.join(HERE, "mousie_sklearn.npz"), X=X, sklearn_score=gmm.score(X),
             sklearn_weights=gmm.weights_, sklearn_means=gmm.means_, sklearn_covariances=gmm.covariances_)
    print("mousie: sklearn score", gmm.score(X), "iters", gmm.n_iter_)


def synth_mixture(rng, n, d, K, sep):
    means = sep * rng.standard_normal((K, d))
    covs = np.empty((K, d, d))
    for k in range(K):
        A = rng.standard_normal((d, d))
        covs[k] = A @ A.T / d + 0.5 * np.eye(d)
    w = rng.uniform(0.5, 1.5, K)
    w /= w.sum()
    comp = rng.choice(K, size=n, p=w)
    X = np.empty((n, d))
    for k in range(K):
        idx = np.where(comp == k)[0]
        L = np.linalg.cholesky(covs[k])
        X[idx] = means[k] + rng.standard_normal((idx.size, d)) @ L.T
    return X, means, covs, w


def make_em_onestep(tag, seed, n, d, K, sep):
    rng = np.random.default_rng(seed)
    X, means, covs, w = synth_mixture(rng, n, d, K, sep)
    # Perturbed starting point (so the step actually moves).
    mu0 = means + 0.3 * rng.standard_normal((K, d))
    Sigma0 = np.empty_like(covs)
    for k in range(K):
        B = rng.standard_normal((d, d)) * 0.1
        Sigma0[k] = covs[k] + B @ B.T + 0.1 * np.eye(d)
    pi0 = rng.uniform(0.5, 1.5, K)
    pi0 /= pi0.sum()

    # Independent E-step (scipy): log N(x | mu0_k, Sigma0_k) + log pi0_k.
    logw = np.stack([scipy.stats.multivariate_normal(mu0[k], Sigma0[k]).logpdf(X) + np.log(pi0[k]) for k in range(K)], axis=1)
    logw = logw.reshape(n, K)
    lse = scipy.special.logsumexp(logw, axis=1)
    R0 = np.exp(logw - lse[:, None])
    ll0 = lse.mean()
    labels0 = np.argmax(R0, axis=1).astype(np.uint32)
    srt = np.sort(R0, axis=1)
    margin = (srt[:, -1] - srt[:, -2]).min() if K > 1 else 1.0

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gmm = sklearn.mixture.GaussianMixture(K, covariance_type="full", tol=0.0, max_iter=1, reg_covar=1e-15,
                                              weights_init=pi0, means_init=mu0,
                                              precisions_init=np.linalg.inv(Sigma0), random_state=0)
        gmm.fit(X)
    assert abs(gmm.lower_bound_ - ll0) < 1e-10 * max(1, abs(ll0)), (gmm.lower_bound_, ll0)
    np.savez(os.path.join(HERE, f"em_onestep_{tag}.npz"), X=X, pi0=pi0, mu0=mu0, Sigma0=Sigma0,
             ll0=ll0, sklearn_lower_bound=gmm.lower_bound_, R0=R0, labels0=labels0, label_margin=margin,
             pi1=gmm.weights_, mu1=gmm.means_, Sigma1=gmm.covariances_)
    print(f"em_onestep_{tag}: n={n} d={d} K={K} ll0={ll0:.12g} min label margin={margin:.3g}")


def make_em_onestep_diag(tag, seed, n, d, K, sep):
    rng = np.random.default_rng(seed)
    means = sep * rng.standard_normal((K, d))
    var = rng.uniform(0.5, 2.0, (K, d))
    w = rng.uniform(0.5, 1.5, K)
    w /= w.sum()
    comp = rng.choice(K, size=n, p=w)
    X = means[comp] + rng.standard_normal((n, d)) * np.sqrt(var[comp])
    mu0 = means + 0.3 * rng.standard_normal((K, d))
    var0 = var * rng.uniform(0.8, 1.3, (K, d)) + 0.05
    pi0 = rng.uniform(0.5, 1.5, K)
    pi0 /= pi0.sum()

    logw = np.stack([scipy.stats.multivariate_normal(mu0[k], np.diag(var0[k])).logpdf(X) + np.log(pi0[k]) for k in range(K)], axis=1)
    logw = logw.reshape(n, K)
    lse = scipy.special.logsumexp(logw, axis=1)
    R0 = np.exp(logw - lse[:, None])
    ll0 = lse.mean()
    labels0 = np.argmax(R0, axis=1).astype(np.uint32)
    srt = np.sort(R0, axis=1)
    margin = (srt[:, -1] - srt[:, -2]).min() if K > 1 else 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gmm = sklearn.mixture.GaussianMixture(K, covariance_type="diag", tol=0.0, max_iter=1, reg_covar=1e-15,
                                              weights_init=pi0, means_init=mu0, precisions_init=1.0 / var0, random_state=0)
        gmm.fit(X)
    assert abs(gmm.lower_bound_ - ll0) < 1e-10 * max(1, abs(ll0)), (gmm.lower_bound_, ll0)
    np.savez(os.path.join(HERE, f"em_onestep_diag_{tag}.npz"), X=X, pi0=pi0, mu0=mu0, var0=var0,
             ll0=ll0, sklearn_lower_bound=gmm.lower_bound_, R0=R0, labels0=labels0, label_margin=margin,
             pi1=gmm.weights_, mu1=gmm.means_, var1=gmm.covariances_)
    print(f"em_onestep_diag_{tag}: n={n} d={d} K={K} ll0={ll0:.12g} min label margin={margin:.3g}")


def make_kmeans_onestep(tag, seed, n, d, K, sep):
    rng = np.random.default_rng(seed)
    X, means, _, _ = synth_mixture(rng, n, d, K, sep)
    C0 = means + 0.3 * rng.standard_normal((K, d))
    # Direct squared distances, summed over dimensions in order (ML/KMeans.cpp:158).
    D = np.zeros((n, K))
    for j in range(d):
        diff = X[:, j][:, None] - C0[:, j][None, :]
        D += diff * diff
    labels0 = np.argmin(D, axis=1).astype(np.uint32)
    inertia0 = D[np.arange(n), labels0].sum()
    srt = np.sort(D, axis=1)
    margin = (srt[:, 1] - srt[:, 0]).min() if K > 1 else 1.0
    counts = np.bincount(labels0, minlength=K)
    C1 = np.zeros((K, d))
    for k in range(K):
        if counts[k]:
            C1[k] = X[labels0 == k].mean(axis=0)   # empty cluster -> origin (ML/KMeans.cpp:184)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = sklearn.cluster.KMeans(K, init=C0, n_init=1, max_iter=1, algorithm="lloyd", tol=0.0).fit(X)
    if counts.min() > 0:
        assert np.allclose(km.cluster_centers_, C1, rtol=1e-12, atol=1e-12)
    np.savez(os.path.join(HERE, f"kmeans_onestep_{tag}.npz"), X=X, C0=C0, labels0=labels0, inertia0=inertia0,
             distance_margin=margin, counts0=counts, C1=C1, sklearn_centers=km.cluster_centers_)
    print(f"kmeans_onestep_{tag}: n={n} d={d} K={K} inertia0={inertia0:.12g} min margin={margin:.3g}")


def make_diag_set():
    make_em_onestep_diag("d4_K3", 31, 800, 4, 3, 2.5)
    make_em_onestep_diag("d16_K16", 32, 1500, 16, 16, 2.0)     # the shape of BASELINE.json configs[1]
    make_em_onestep_diag("d32_K8", 33, 1000, 32, 8, 1.5)
    make_em_onestep_diag("d7_K40", 34, 1200, 7, 40, 2.5)


if __name__ == "__main__":
    import sys
    if sys.argv[1:] == ["diag"]:       # only the diagonal fixtures (added in round 2; the others are left untouched)
        make_diag_set()
        sys.exit(0)
    make_mousie()
    make_em_onestep("d2_K3", 11, 500, 2, 3, 3.0)
    make_em_onestep("d3_K2", 12, 400, 3, 2, 3.0)
    make_em_onestep("d3_K1", 13, 300, 3, 1, 3.0)
    make_em_onestep("d4_K3", 14, 1000, 4, 3, 2.5)
    make_em_onestep("d16_K16", 15, 1500, 16, 16, 2.0)
    make_em_onestep("d32_K16", 16, 1200, 32, 16, 2.0)
    make_em_onestep("d13_K5", 17, 600, 13, 5, 2.0)    # just below the reference's d<14 / d<15 switches
    make_em_onestep("d15_K5", 18, 600, 15, 5, 2.0)    # just above
    make_kmeans_onestep("d2_K3", 21, 500, 2, 3, 3.0)
    make_kmeans_onestep("d8_K32", 22, 2000, 8, 32, 2.0)
    make_kmeans_onestep("d3_K1", 23, 200, 3, 1, 2.0)
    make_diag_set()
