"""LinearRegression::calculate_XXt_beta (ML/LinearRegression.cpp:201-230). The oracle restatement
is pinned against numpy on CPU; the product (X X^T and X y on the GPU) is checked against the oracle on the GPU box."""
import numpy as np
import pytest


def _problem(n, q, seed):
    rng = np.random.default_rng(seed)
    X = np.ascontiguousarray(rng.standard_normal((n, q)) * rng.uniform(0.5, 3, q) + rng.uniform(-2, 2, q))
    beta = rng.standard_normal(q)
    y = X @ beta + 0.1 * rng.standard_normal(n)
    return X, y


@pytest.mark.parametrize("n,q,ridge", [(50, 3, 0.0), (400, 16, 0.5), (1000, 32, 0.0)])
def test_oracle_matches_numpy(oracle, n, q, ridge):
    X, y = _problem(n, q, n + q)
    lam = np.full(q, ridge)
    XXt, beta = oracle.calculate_XXt_beta(X, y, lam)
    A = X.T @ X + np.diag(lam)
    assert np.max(np.abs(XXt - A)) <= 1e-13 * np.max(np.abs(A))
    assert np.max(np.abs(beta - np.linalg.solve(A, X.T @ y))) <= 1e-10 * np.max(np.abs(beta))


def test_argument_errors(oracle):
    from ml_amd import _lib
    X, y = _problem(20, 3, 1)
    for fn in (oracle.calculate_XXt_beta, _lib.calculate_XXt_beta):
        with pytest.raises(Exception, match="cannot be negative"):
            fn(X, y, np.array([0.0, -1.0, 0.0]))
        with pytest.raises(Exception, match="same size as the number of features"):
            fn(X, y, np.zeros(2))
        with pytest.raises(Exception, match="different number of data points"):
            fn(X, y[:-1], np.zeros(3))
        with pytest.raises(Exception, match="Not enough data points"):
            fn(X[:2], y[:2], np.zeros(3))


@pytest.mark.gpu
@pytest.mark.parametrize("n,q,ridge", [(50, 3, 0.0), (4000, 16, 0.5), (100000, 32, 0.0), (300001, 7, 2.0)])
def test_product_matches_oracle(oracle, n, q, ridge):
    from ml_amd import _lib
    X, y = _problem(n, q, n + q)
    lam = np.full(q, ridge)
    XXt, beta = _lib.calculate_XXt_beta(X, y, lam)
    XXt_ref, beta_ref = oracle.calculate_XXt_beta(X, y, lam)
    assert np.max(np.abs(XXt - XXt_ref)) <= 1e-12 * np.max(np.abs(XXt_ref))
    assert np.max(np.abs(beta - beta_ref)) <= 1e-9 * np.max(np.abs(beta_ref))
    assert np.array_equal(XXt, XXt.T)
