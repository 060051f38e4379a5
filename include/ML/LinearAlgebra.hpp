#pragma once
/* Same helpers, names and error behaviour as the reference's ML/LinearAlgebra.hpp:9-32. They are host-side
 * single-vector utilities; inside EM their work is done by the HIP kernels (ml_amd/csrc/device). */
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
}
}
