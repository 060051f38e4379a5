// Host-side helpers with the interface and behaviour of the reference's ML/LinearAlgebra.cpp:8-73.
#include "ML/LinearAlgebra.hpp"

#include <stdexcept>

namespace ml {
namespace LinearAlgebra {

double xAx_symmetric(const MatrixXd& A, ConstVectorRef x)
{
    if (A.rows() != A.cols()) throw std::invalid_argument("A matrix is not square");
    if (x.size() != A.rows()) throw std::invalid_argument("x has wrong size");
    const Index n = A.rows();
    if (n >= 15) {
        // Large sizes: y = sym(A) x from the upper triangle, then x . y (the reference switches to an Eigen
        // selfadjointView product here, ML/LinearAlgebra.cpp:29); the two-level sum also keeps the rounding error
        // at the 1e-14 relative level the reference's test asks for at n = 1024.
        double total = 0;
        for (Index r = 0; r < n; ++r) {
            double y = 0;
            for (Index c = 0; c < n; ++c) y += (c >= r ? A(r, c) : A(c, r)) * x[c];
            total += x[r] * y;
        }
        return total;
    }
    // Upper triangle only; column by column: diagonal term, then the doubled off-diagonal terms above it.
    double sum = 0;
    for (Index c = 0; c < n; ++c) {
        const double xc = x[c];
        const double* column = A.col(c);
        sum += column[c] * xc * xc;
        for (Index r = 0; r < c; ++r) sum += 2 * column[r] * xc * x[r];
    }
    return sum;
}

void xxT(const VectorXd& x, MatrixXd& dest)
{
    const Index n = x.size();
    if (dest.rows() != n || dest.cols() != n) dest.resize(n, n);
    for (Index c = 0; c < n; ++c) {
        for (Index r = 0; r < c; ++r) {
            const double v = x[c] * x[r];
            dest(c, r) = v;
            dest(r, c) = v;
        }
        dest(c, c) = x[c] * x[c];
    }
}

void add_a_xxT(const VectorXd& x, MatrixXd& dest, const double a)
{
    const Index n = x.size();
    if (dest.rows() != n || dest.cols() != n) throw std::invalid_argument("Expected square matrix with the same size as x");
    for (Index c = 0; c < n; ++c) {
        const double axc = a * x[c];
        for (Index r = 0; r < c; ++r) {
            const double v = axc * x[r];
            dest(c, r) += v;
            dest(r, c) += v;
        }
        dest(c, c) += axc * x[c];
    }
}

}  // namespace LinearAlgebra
}  // namespace ml
