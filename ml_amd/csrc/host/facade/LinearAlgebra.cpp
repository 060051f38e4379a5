// Host-side helpers with the interface and behaviour of the reference's ML/LinearAlgebra.cpp:8-73.
#include "ML/LinearAlgebra.hpp"

#include <stdexcept>

namespace ml {
namespace LinearAlgebra {

double xAx_symmetric(const double* A, Index rows, Index cols, Index ld, const double* x, Index x_size)
{
    if (rows != cols) throw std::invalid_argument("A matrix is not square");
    if (x_size != rows) throw std::invalid_argument("x has wrong size");
    const Index n = rows;
    if (n >= 15) {
        // Large sizes: y = sym(A) x from the upper triangle, then x . y (the reference switches to an Eigen
        // selfadjointView product here, ML/LinearAlgebra.cpp:29); the two-level sum also keeps the rounding error
        // at the 1e-14 relative level the reference's test asks for at n = 1024.
        double total = 0;
        for (Index r = 0; r < n; ++r) {
            double y = 0;
            for (Index c = 0; c < n; ++c) y += (c >= r ? A[c * ld + r] : A[r * ld + c]) * x[c];
            total += x[r] * y;
        }
        return total;
    }
    // Upper triangle only; column by column: diagonal term, then the doubled off-diagonal terms above it.
    double sum = 0;
    for (Index c = 0; c < n; ++c) {
        const double xc = x[c];
        const double* column = A + c * ld;
        sum += column[c] * xc * xc;
        for (Index r = 0; r < c; ++r) sum += 2 * column[r] * xc * x[r];
    }
    return sum;
}

void xxT(const double* x, Index n, double* dest, Index ld)
{
    for (Index c = 0; c < n; ++c) {
        for (Index r = 0; r < c; ++r) {
            const double v = x[c] * x[r];
            dest[r * ld + c] = v;
            dest[c * ld + r] = v;
        }
        dest[c * ld + c] = x[c] * x[c];
    }
}

void add_a_xxT(const double* x, Index n, double* dest, Index dest_rows, Index dest_cols, Index ld, const double a)
{
    if (dest_rows != n || dest_cols != n) throw std::invalid_argument("Expected square matrix with the same size as x");
    for (Index c = 0; c < n; ++c) {
        const double axc = a * x[c];
        for (Index r = 0; r < c; ++r) {
            const double v = axc * x[r];
            dest[r * ld + c] += v;
            dest[c * ld + r] += v;
        }
        dest[c * ld + c] += axc * x[c];
    }
}

double xAx_symmetric(const MatrixXd& A, ConstVectorRef x)
{
    return xAx_symmetric(A.data(), A.rows(), A.cols(), A.rows(), x.data(), x.size());
}

void xxT(const VectorXd& x, MatrixXd& dest)
{
    const Index n = x.size();
    if (dest.rows() != n || dest.cols() != n) dest.resize(n, n);
    xxT(x.data(), n, dest.data(), n);
}

void add_a_xxT(const VectorXd& x, MatrixXd& dest, const double a)
{
    add_a_xxT(x.data(), x.size(), dest.data(), dest.rows(), dest.cols(), dest.rows(), a);
}

}  // namespace LinearAlgebra
}  // namespace ml
