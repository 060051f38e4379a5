// ml::LinearRegression::calculate_XXt_beta with the behaviour of the reference's ML/LinearRegression.cpp:201-230:
// b = X y and X X^T on the GPU (one pass over the resident block), ridge on the diagonal; the q x q pivoted LDL^T
// factorisation (what Eigen::LDLT computes there, :228) and the solve (:229) on the host.
#include "ML/LinearRegression.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>

#include "ML/Device.hpp"
#include "mlhip.h"

namespace ml {

// Robust Cholesky with diagonal pivoting, unblocked, in place on the lower triangle (the algorithm Eigen documents for
// LDLT: at step k the largest remaining |diagonal| entry is swapped into position k, then column k of L is formed).
LDLT& LDLT::compute(ConstMatrixRef A)
{
    if (A.rows() != A.cols()) throw std::invalid_argument("LDLT: matrix is not square");
    const Index n = A.rows();
    ldlt_.resize(n, n);
    for (Index j = 0; j < n; ++j)
        for (Index i = 0; i < n; ++i) ldlt_(i, j) = i >= j ? A(i, j) : A(j, i);     // symmetric fill from the lower triangle
    transpositions_.assign(static_cast<std::size_t>(n), 0);
    sign_ = 0;
    MatrixXd& m = ldlt_;
    std::vector<double> temp(static_cast<std::size_t>(n));
    for (Index k = 0; k < n; ++k) {
        Index p = k;
        double biggest = std::abs(m(k, k));
        for (Index i = k + 1; i < n; ++i)
            if (std::abs(m(i, i)) > biggest) { biggest = std::abs(m(i, i)); p = i; }
        transpositions_[static_cast<std::size_t>(k)] = p;
        if (p != k) {
            // symmetric swap of rows/columns k and p, touching the lower triangle only
            for (Index j = 0; j < k; ++j) std::swap(m(k, j), m(p, j));
            for (Index i = p + 1; i < n; ++i) std::swap(m(i, k), m(i, p));
            std::swap(m(k, k), m(p, p));
            for (Index i = k + 1; i < p; ++i) std::swap(m(i, k), m(p, i));
        }
        // m(k,k) -= sum_j L(k,j)^2 D_j ; column k below the diagonal -= L(i, :k) (D .* L(k, :k)), then / D_k
        for (Index j = 0; j < k; ++j) temp[static_cast<std::size_t>(j)] = m(j, j) * m(k, j);
        double dk = m(k, k);
        for (Index j = 0; j < k; ++j) dk -= m(k, j) * temp[static_cast<std::size_t>(j)];
        m(k, k) = dk;
        for (Index i = k + 1; i < n; ++i) {
            double t = m(i, k);
            for (Index j = 0; j < k; ++j) t -= m(i, j) * temp[static_cast<std::size_t>(j)];
            m(i, k) = t;
        }
        const bool pivot_is_zero = !(std::abs(dk) > 0);
        if (!pivot_is_zero)
            for (Index i = k + 1; i < n; ++i) m(i, k) /= dk;
        else
            for (Index i = k + 1; i < n; ++i) m(i, k) = 0;           // null direction: nothing below it is used
        if (dk > 0) sign_ = (sign_ == 0 || sign_ == 1) ? 1 : 2;
        else if (dk < 0) sign_ = (sign_ == 0 || sign_ == -1) ? -1 : 2;
    }
    return *this;
}

VectorXd LDLT::solve(ConstVectorRef b) const
{
    const Index n = ldlt_.rows();
    if (b.size() != n) throw std::invalid_argument("LDLT: right-hand side has the wrong size");
    VectorXd x(n);
    for (Index i = 0; i < n; ++i) x[i] = b[i];
    for (Index k = 0; k < n; ++k) std::swap(x[k], x[transpositions_[static_cast<std::size_t>(k)]]);       // P b
    for (Index i = 0; i < n; ++i) {                                                                         // L^-1
        double t = x[i];
        for (Index j = 0; j < i; ++j) t -= ldlt_(i, j) * x[j];
        x[i] = t;
    }
    const double tolerance = std::numeric_limits<double>::min();                                            // D^-1 (pseudo-inverse)
    for (Index i = 0; i < n; ++i) x[i] = std::abs(ldlt_(i, i)) > tolerance ? x[i] / ldlt_(i, i) : 0.0;
    for (Index i = n - 1; i >= 0; --i) {                                                                    // L^-T
        double t = x[i];
        for (Index j = i + 1; j < n; ++j) t -= ldlt_(j, i) * x[j];
        x[i] = t;
    }
    for (Index k = n - 1; k >= 0; --k) std::swap(x[k], x[transpositions_[static_cast<std::size_t>(k)]]);   // P^T
    return x;
}

VectorXd LDLT::vectorD() const
{
    VectorXd dvec(ldlt_.rows());
    for (Index i = 0; i < ldlt_.rows(); ++i) dvec[i] = ldlt_(i, i);
    return dvec;
}

MatrixXd LDLT::reconstructedMatrix() const
{
    const Index n = ldlt_.rows();
    MatrixXd r(n, n);
    for (Index j = 0; j < n; ++j)
        for (Index i = 0; i < n; ++i) {
            double t = 0;
            for (Index l = 0; l <= std::min(i, j); ++l) t += (l == i ? 1.0 : ldlt_(i, l)) * ldlt_(l, l) * (l == j ? 1.0 : ldlt_(j, l));
            r(i, j) = t;
        }
    for (Index k = n - 1; k >= 0; --k) {                          // undo the pivoting: P^T (.) P
        const Index p = transpositions_[static_cast<std::size_t>(k)];
        if (p == k) continue;
        for (Index j = 0; j < n; ++j) std::swap(r(k, j), r(p, j));
        for (Index i = 0; i < n; ++i) std::swap(r(i, k), r(i, p));
    }
    return r;
}

namespace LinearRegression {

void calculate_XXt_b(ConstMatrixRef X, ConstVectorRef y, MatrixRef XXt, VectorRef b, ConstVectorRef lambda)
{
    const Index n = X.cols(), q = X.rows();
    double min_lambda = 0;
    for (Index i = 0; i < lambda.size(); ++i) min_lambda = i ? std::min(min_lambda, lambda[i]) : lambda[i];
    if (lambda.size() && min_lambda < 0) throw std::domain_error("Ridge regularisation constant cannot be negative");
    if (lambda.size() != q) throw std::invalid_argument("Lambda vector must have the same size as the number of features");
    if (n != y.size()) throw std::invalid_argument("X matrix has different number of data points than Y has values");
    if (XXt.rows() != q || XXt.cols() != q) throw std::invalid_argument("XXt must be q x q");
    if (b.size() != q) throw std::invalid_argument("b must have q entries");

    // "Not enough data points" is about the whole sample: in a row-sharded job the local block may be shorter than q.
    int world = 1;
    mlhip_ctx* ctx = device::peek_context();
    if (ctx) device::check(mlhip_ctx_world(ctx, &world, nullptr));
    if (world == 1 && n < q) throw std::invalid_argument("Not enough data points for regression");
    if (!ctx || world == 1) ctx = device::context();
    mlhip_data* dev = nullptr;
    device::check(mlhip_data_upload(ctx, X.data(), static_cast<uint32_t>(q), static_cast<uint64_t>(n), X.outerStride(), &dev));
    uint64_t n_global = 0;
    int rc = mlhip_data_shape(dev, nullptr, nullptr, &n_global);
    if (rc == MLHIP_OK && n_global < static_cast<uint64_t>(q)) {
        mlhip_data_free(dev);
        throw std::invalid_argument("Not enough data points for regression");
    }
    std::vector<double> xxt(static_cast<std::size_t>(q * q));
    if (rc == MLHIP_OK) rc = mlhip_xxt_xy(ctx, dev, y.data(), xxt.data(), b.data());
    mlhip_data_free(dev);
    device::check(rc);
    // `if (lambda.minCoeff())` in the reference (:221): the ridge is added only when its smallest entry is non-zero.
    const bool ridge = min_lambda != 0;
    for (Index j = 0; j < q; ++j)
        for (Index i = 0; i < q; ++i) XXt(i, j) = xxt[static_cast<std::size_t>(j * q + i)] + (ridge && i == j ? lambda[i] : 0.0);
}

VectorXd calculate_XXt_beta(ConstMatrixRef X, ConstVectorRef y, MatrixRef XXt, LDLT& xxt_decomp, ConstVectorRef lambda)
{
    VectorXd b(X.rows());
    calculate_XXt_b(X, y, XXt, b, lambda);
    xxt_decomp.compute(ConstMatrixRef(XXt.data(), XXt.rows(), XXt.cols(), XXt.outerStride()));   // :228
    return xxt_decomp.solve(b);                                                                    // :229
}

}  // namespace LinearRegression
}  // namespace ml
