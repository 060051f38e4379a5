#pragma once
/* Same helpers, names and error behaviour as the reference's ML/LinearAlgebra.hpp:9-32. They are host-side
 * single-vector utilities; inside EM their work is done by the HIP kernels (ml_amd/csrc/device).
 * The library is built without Eigen, so the exported symbols take the types of Dense.hpp or raw column-major buffers; where
 * <Eigen/Core> is available the reference's exact Eigen-typed signatures exist as inline forwards (see the end of this file;
 * parsed against tests/cpp/eigen_shim only -- real Eigen is absent from this repository's build environment). */
#include "Dense.hpp"
#include "dll.hpp"

namespace ml {
namespace LinearAlgebra {
/** x^T A x for a symmetric matrix A; only the upper triangle of A is read.
@throw std::invalid_argument If `A` is not square or `x.size() != A.rows()`. */
DLL_DECLSPEC double xAx_symmetric(const MatrixXd& A, ConstVectorRef x);
/** dest = x x^T (dest is resized if necessary). */
DLL_DECLSPEC void xxT(const VectorXd& x, MatrixXd& dest);
/** dest += a x x^T.
@throw std::invalid_argument If `dest` is not square with the size of `x`. */
DLL_DECLSPEC void add_a_xxT(const VectorXd& x, MatrixXd& dest, double a);

/** The same three on raw column-major storage (`ld` = distance in doubles between columns); same checks, same arithmetic. */
DLL_DECLSPEC double xAx_symmetric(const double* A, Index rows, Index cols, Index ld, const double* x, Index x_size);
DLL_DECLSPEC void xxT(const double* x, Index n, double* dest, Index ld);
DLL_DECLSPEC void add_a_xxT(const double* x, Index n, double* dest, Index dest_rows, Index dest_cols, Index ld, double a);

#ifdef MLHIP_HAVE_EIGEN
/* The reference's signatures (reference ML/LinearAlgebra.hpp:18, 24, 31), forwarding without copies. */
inline double xAx_symmetric(const Eigen::MatrixXd& A, Eigen::Ref<const Eigen::VectorXd> x)
{
    return xAx_symmetric(A.data(), A.rows(), A.cols(), A.rows(), x.data(), x.size());
}
inline void xxT(const Eigen::VectorXd& x, Eigen::MatrixXd& dest)
{
    if (dest.rows() != x.size() || dest.cols() != x.size()) dest.resize(x.size(), x.size());
    xxT(x.data(), x.size(), dest.data(), dest.rows());
}
inline void add_a_xxT(const Eigen::VectorXd& x, Eigen::MatrixXd& dest, double a)
{
    add_a_xxT(x.data(), x.size(), dest.data(), dest.rows(), dest.cols(), dest.rows(), a);
}
#endif
}
}
