// ml::LinearRegression::calculate_XXt_beta with the behaviour of the reference's ML/LinearRegression.cpp:201-230:
// b = X y and X X^T on the GPU (one pass over the resident block), ridge on the diagonal, q x q solve on the host.
#include "ML/LinearRegression.hpp"

#include <cmath>
#include <stdexcept>
#include <vector>

#include "ML/Device.hpp"
#include "mlhip.h"

namespace ml {
namespace LinearRegression {

VectorXd calculate_XXt_beta(ConstMatrixRef X, ConstVectorRef y, MatrixRef XXt, ConstVectorRef lambda)
{
    const Index n = X.cols(), q = X.rows();
    double min_lambda = 0;
    for (Index i = 0; i < lambda.size(); ++i) min_lambda = i ? std::min(min_lambda, lambda[i]) : lambda[i];
    if (lambda.size() && min_lambda < 0) throw std::domain_error("Ridge regularisation constant cannot be negative");
    if (lambda.size() != q) throw std::invalid_argument("Lambda vector must have the same size as the number of features");
    if (n != y.size()) throw std::invalid_argument("X matrix has different number of data points than Y has values");
    if (n < q) throw std::invalid_argument("Not enough data points for regression");
    if (XXt.rows() != q || XXt.cols() != q) throw std::invalid_argument("XXt must be q x q");

    mlhip_ctx* ctx = device::context();
    mlhip_data* dev = nullptr;
    device::check(mlhip_data_upload(ctx, X.data(), static_cast<uint32_t>(q), static_cast<uint64_t>(n), X.outerStride(), &dev));
    std::vector<double> xxt(static_cast<std::size_t>(q * q)), b(static_cast<std::size_t>(q));
    const int rc = mlhip_xxt_xy(ctx, dev, y.data(), xxt.data(), b.data());
    mlhip_data_free(dev);
    device::check(rc);
    // `if (lambda.minCoeff())` in the reference (:221): the ridge is added only when its smallest entry is non-zero.
    const bool ridge = min_lambda != 0;
    for (Index j = 0; j < q; ++j)
        for (Index i = 0; i < q; ++i) XXt(i, j) = xxt[static_cast<std::size_t>(j * q + i)] + (ridge && i == j ? lambda[i] : 0.0);

    // L D L^T factorisation (no square roots) and the two triangular solves.
    std::vector<double> L(static_cast<std::size_t>(q * q), 0.0), D(static_cast<std::size_t>(q));
    for (Index j = 0; j < q; ++j) {
        double dj = XXt(j, j);
        for (Index l = 0; l < j; ++l) dj -= L[l * q + j] * L[l * q + j] * D[l];
        if (!(dj > 0)) throw std::runtime_error("calculate_XXt_beta: X X^T + diag(lambda) is not positive definite");
        D[j] = dj;
        L[j * q + j] = 1.0;
        for (Index i = j + 1; i < q; ++i) {
            double t = XXt(i, j);
            for (Index l = 0; l < j; ++l) t -= L[l * q + i] * L[l * q + j] * D[l];
            L[j * q + i] = t / dj;
        }
    }
    VectorXd beta(q);
    for (Index i = 0; i < q; ++i) {
        double t = b[i];
        for (Index l = 0; l < i; ++l) t -= L[l * q + i] * beta[l];
        beta[i] = t;
    }
    for (Index i = 0; i < q; ++i) beta[i] /= D[i];
    for (Index i = q - 1; i >= 0; --i) {
        double t = beta[i];
        for (Index l = i + 1; l < q; ++l) t -= L[i * q + l] * beta[l];
        beta[i] = t;
    }
    return beta;
}

}  // namespace LinearRegression
}  // namespace ml
