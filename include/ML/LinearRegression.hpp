#pragma once
/* The dense helper of the reference's linear-regression code that shares the covariance contraction of the EM path
 * (SURVEY.md section 8, row f4): ml::LinearRegression::calculate_XXt_beta, reference ML/LinearRegression.hpp:412,
 * ML/LinearRegression.cpp:201-230 -- the reference's 5-argument shape, decomposition out-parameter included (ols / ridge
 * reuse it for the coefficient covariance). X X^T and X y are formed on the GPU in one pass (mlhip_xxt_xy); the q x q
 * factorisation and solve stay on the host. The rest of the reference's LinearRegression namespace is out of scope. */
#include <vector>

#include "Dense.hpp"
#include "dll.hpp"

#if defined(MLHIP_HAVE_EIGEN) && __has_include(<Eigen/Cholesky>)
#include <Eigen/Cholesky>
#define MLHIP_HAVE_EIGEN_CHOLESKY 1
#endif

namespace ml {

/** Stands in for Eigen::LDLT<Eigen::MatrixXd> in signatures (the library is built without Eigen): the robust Cholesky
decomposition with diagonal pivoting of a symmetric positive or negative SEMI-definite matrix, P A P^T = L D L^T (L unit lower
triangular). Like Eigen's, solve() treats pivots with |D_ii| <= the smallest normal double as zero (minimum-norm behaviour on
the null space), so collinear features with lambda = 0 still give a solution. */
class LDLT {
public:
    LDLT() = default;
    explicit LDLT(ConstMatrixRef A) { compute(A); }
    /** Factorises the symmetric matrix whose LOWER triangle is stored in `A`. @throw std::invalid_argument If not square. */
    DLL_DECLSPEC LDLT& compute(ConstMatrixRef A);
    /** x with A x = b. @throw std::invalid_argument On a size mismatch. */
    DLL_DECLSPEC VectorXd solve(ConstVectorRef b) const;
    Index rows() const { return ldlt_.rows(); }
    Index cols() const { return ldlt_.cols(); }
    /** L (strict lower triangle, unit diagonal implied) and D (diagonal) packed in one matrix, in PIVOTED order. */
    const MatrixXd& matrixLDLT() const { return ldlt_; }
    /** transpositions()[k] = row/column swapped with k at step k (P = product of these swaps, applied in order). */
    const std::vector<Index>& transpositions() const { return transpositions_; }
    /** D as a vector. */
    DLL_DECLSPEC VectorXd vectorD() const;
    bool isPositive() const { return sign_ == 0 || sign_ == 1; }    // no negative pivot seen (semi-definite counts), as Eigen::LDLT
    bool isNegative() const { return sign_ == 0 || sign_ == -1; }   // an indefinite matrix (sign_ == 2) is neither
    /** A reconstructed from the factors (P^T L D L^T P), e.g. for tests. */
    DLL_DECLSPEC MatrixXd reconstructedMatrix() const;
private:
    MatrixXd ldlt_;
    std::vector<Index> transpositions_;
    int sign_ = 0;     // 0: all pivots zero so far, +1 / -1: definite sign seen, 2: indefinite (both signs)
public:
    bool isIndefinite() const { return sign_ == 2; }
};

namespace LinearRegression {
/** Solves (X X^T + diag(lambda)) beta = X y and leaves the decomposition of the regularised matrix in `xxt_decomp`
(reference ML/LinearRegression.cpp:201-230).
@param[in] X q x N, one data point per column (this rank's columns in a row-sharded job). @param[in] y N targets.
@param[out] XXt q x q, receives X X^T + diag(lambda). @param[out] xxt_decomp its LDL^T decomposition.
@param[in] lambda q non-negative ridge constants.
@throw std::domain_error If a lambda is negative. @throw std::invalid_argument On size mismatches or fewer data points
(over all ranks) than features. @throw std::runtime_error On device failures. */
DLL_DECLSPEC VectorXd calculate_XXt_beta(ConstMatrixRef X, ConstVectorRef y, MatrixRef XXt, LDLT& xxt_decomp, ConstVectorRef lambda);

/** Device part only: XXt = X X^T + diag(lambda) (q x q) and b = X y (q), after the reference's argument checks. */
DLL_DECLSPEC void calculate_XXt_b(ConstMatrixRef X, ConstVectorRef y, MatrixRef XXt, VectorRef b, ConstVectorRef lambda);

#ifdef MLHIP_HAVE_EIGEN_CHOLESKY
/** The reference's exact signature: the contraction on the GPU, decomposition and solve by the caller's Eigen::LDLT object. */
inline Eigen::VectorXd calculate_XXt_beta(const Eigen::Ref<const Eigen::MatrixXd> X, const Eigen::Ref<const Eigen::VectorXd> y,
                                          Eigen::Ref<Eigen::MatrixXd> XXt, Eigen::LDLT<Eigen::MatrixXd>& xxt_decomp,
                                          const Eigen::Ref<const Eigen::VectorXd> lambda)
{
    Eigen::VectorXd b(X.rows());
    calculate_XXt_b(ConstMatrixRef(X), ConstVectorRef(y), MatrixRef(XXt), VectorRef(b), ConstVectorRef(lambda));
    xxt_decomp.compute(XXt);
    return xxt_decomp.solve(b);
}
#endif
}
}
