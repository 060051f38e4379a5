"""Pins the CPU restatement (oracle/) against the committed third-party golden vectors
(tests/golden/make_golden.py: scikit-learn / scipy). CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

DIAG_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "em_onestep_diag_*.npz")))
EM_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "em_onestep_*.npz"))
                  if os.path.basename(p) not in DIAG_CASES)
KM_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kmeans_onestep_*.npz")))

# Tolerance policy (DESIGN.md "Tolerances"): per-step fixtures agree to 1e-12 relative.
RTOL = 1e-12


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.mark.parametrize("case", EM_CASES)
def test_em_one_step_matches_sklearn(oracle, case):
    g = load_golden(case)
    X = g["X"]
    n, d = X.shape
    K = g["pi0"].size
    em = oracle.EM(K)
    em.set_parameters(g["mu0"], g["Sigma0"], g["pi0"])
    em.expectation_step(X)
    assert abs(em.log_likelihood - float(g["ll0"])) <= RTOL * abs(float(g["ll0"]))
    R = em.responsibilities
    assert np.max(np.abs(R - g["R0"])) < 1e-12
    em.calculate_labels()
    assert np.array_equal(em.labels, g["labels0"])
    em.maximisation_step(X)
    assert relerr(em.mixing_probabilities, g["pi1"]) < RTOL
    assert relerr(em.means, g["mu1"]) < RTOL
    # sklearn's covariances are exactly symmetric products; ours accumulate rank-1 updates.
    assert relerr(em.covariances, g["Sigma1"]) < 1e-11


@pytest.mark.parametrize("case", DIAG_CASES)
def test_diagonal_em_one_step_matches_sklearn(oracle, case):
    """The diagonal-covariance extension of the oracle (BASELINE.json configs[1]; no reference counterpart) against
    scikit-learn covariance_type='diag' / scipy."""
    g = load_golden(case)
    X = g["X"]
    K, d = g["mu0"].shape
    em = oracle.EM(K)
    em.set_covariance_type("diag")
    em.set_parameters(g["mu0"], np.stack([np.diag(v) for v in g["var0"]]), g["pi0"])
    em.expectation_step(X)
    assert abs(em.log_likelihood - float(g["ll0"])) <= RTOL * abs(float(g["ll0"]))
    assert np.max(np.abs(em.responsibilities - g["R0"])) < 1e-12
    em.calculate_labels()
    assert np.array_equal(em.labels, g["labels0"])
    em.maximisation_step(X)
    assert relerr(em.mixing_probabilities, g["pi1"]) < RTOL
    assert relerr(em.means, g["mu1"]) < RTOL
    S = em.covariances
    assert relerr(np.stack([np.diag(S[k]) for k in range(K)]), g["var1"]) < 1e-11
    off = S - np.stack([np.diag(np.diag(S[k])) for k in range(K)])
    assert np.all(off == 0)                                  # off-diagonal entries stay exactly 0


def test_diagonal_em_fit_converges_to_the_diagonal_of_its_own_fixed_point(oracle):
    """A full fit in diagonal mode: monotone log-likelihood is implied by convergence; parameters are a fixed point of
    one more diagonal E+M step to the fit's tolerance."""
    g = load_golden("em_onestep_diag_d4_K3.npz")
    X = g["X"]
    K, d = g["mu0"].shape
    em = oracle.EM(K)
    em.set_covariance_type("diag")
    em.set_means_initialiser(oracle.FIXED, g["mu0"])
    em.set_absolute_tolerance(1e-12)
    em.set_relative_tolerance(1e-12)
    assert em.fit(X)
    mu, S, pi, ll = em.means.copy(), em.covariances.copy(), em.mixing_probabilities.copy(), em.log_likelihood
    again = oracle.EM(K)
    again.set_covariance_type("diag")
    again.set_parameters(mu, S, pi)
    again.expectation_step(X)
    again.maximisation_step(X)
    assert abs(again.log_likelihood - ll) < 1e-8 * abs(ll)
    assert relerr(again.means, mu) < 1e-5 and relerr(again.covariances, S) < 1e-5


@pytest.mark.parametrize("case", KM_CASES)
def test_kmeans_one_step_matches_numpy_sklearn(oracle, case):
    g = load_golden(case)
    X = g["X"]
    n, d = X.shape
    K = g["C0"].shape[0]
    km = oracle.KMeans(K)
    km.set_centroids(g["C0"], n)
    km.assignment_step(X)
    assert np.array_equal(km.labels, g["labels0"])
    assert abs(km.inertia - float(g["inertia0"])) <= 1e-13 * float(g["inertia0"])
    km.update_step(X)
    assert relerr(km.centroids, g["C1"]) < 1e-13


def test_mousie_em_matches_sklearn_score(oracle):
    """The reference's own Python test (cppyml/tests/test_clustering.py:47-74)."""
    g = load_golden("mousie_sklearn.npz")
    X = g["X"]
    em = oracle.EM(3)
    em.set_seed(42)
    em.set_absolute_tolerance(1e-10)
    em.set_relative_tolerance(0)
    em.set_means_initialiser(oracle.KPP)
    em.set_maximum_steps(1000)
    assert em.fit(X)
    assert abs(em.log_likelihood - float(g["sklearn_score"])) < 1e-10
    u = em.assign_responsibilities(np.array([0.0, 0.0]))
    assert len(u) == 3
    assert abs(sum(u) - 1) <= 1e-15
    assert min(u) >= 0
    assert abs(max(u) - 1) < 1e-9


def test_mousie_kmeans_reference_python_test(oracle):
    """cppyml/tests/test_clustering.py:76-95."""
    g = load_golden("mousie_sklearn.npz")
    X = g["X"]
    km = oracle.KMeans(3)
    km.set_seed(42)
    km.set_absolute_tolerance(1e-10)
    km.set_centroids_initialiser(oracle.KPP)
    km.set_maximum_steps(1000)
    km.set_number_initialisations(10)
    assert km.fit(X)
    assert km.inertia > 0
    labels = km.labels
    assert labels.min() == 0 and labels.max() == 2
    C = km.centroids
    assert C.shape == (3, 2)
    for i, c in enumerate(C):
        label, dist = km.assign_label(c)
        assert label == i and dist == 0
