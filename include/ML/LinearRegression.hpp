#pragma once
/* The dense helper of the reference's linear-regression code that shares the covariance contraction of the EM path
 * (SURVEY.md section 8, row f4): ml::LinearRegression::calculate_XXt_beta, ML/LinearRegression.hpp:412,
 * ML/LinearRegression.cpp:201-230. X X^T and X y are formed on the GPU in one pass (mlhip_xxt_xy); the q x q solve
 * stays on the host. The rest of the reference's LinearRegression namespace is out of scope. */
#include "Dense.hpp"
#include "dll.hpp"

namespace ml {
namespace LinearRegression {
/** Solves (X X^T + diag(lambda)) beta = X y.
@param[in] X q x N, one data point per column. @param[in] y N targets. @param[out] XXt q x q, receives X X^T + diag(lambda).
@param[in] lambda q non-negative ridge constants.
@throw std::domain_error If a lambda is negative. @throw std::invalid_argument On size mismatches or N < q.
@throw std::runtime_error If the regularised matrix is not positive definite, or on device failures. */
DLL_DECLSPEC VectorXd calculate_XXt_beta(ConstMatrixRef X, ConstVectorRef y, MatrixRef XXt, ConstVectorRef lambda);
}
}
