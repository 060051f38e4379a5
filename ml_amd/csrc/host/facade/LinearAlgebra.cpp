// Host-side helpers with the interface and behaviour of the reference's ML/LinearAlgebra.cpp:8-73.
#include "ML/LinearAlgebra.hpp"

#include <cstddef>
#include <stdexcept>
#include <vector>

namespace ml {
namespace LinearAlgebra {

double xAx_symmetric(const double* A, Index rows, Index cols, Index ld, const double* x, Index x_size)
{
    if (rows != cols) throw std::invalid_argument("A matrix is not square");
    if (x_size != rows) throw std::invalid_argument("x has wrong size");
    const Index n = rows;
    if (n >= 15) {
        // Large sizes: y = sym(A) x from the upper triangle, then x . y -- the reference switches to an Eigen selfadjointView
        // product here (ML/LinearAlgebra.cpp:29). Column c of the upper triangle is contiguous: its part above the diagonal feeds
        // y[0 .. c) (an axpy) and, by symmetry, y[c] (a dot product), so the triangle is read once, with unit stride (round 4
        // walked the mirrored entries with stride ld: 2.2 ms at n = 1024, Benchmarks/bm_LinearAlgebra.cpp). Sums of at most n terms
        // each, then one of n: the rounding error stays at the 1e-14 relative level the reference's test asks for at n = 1024.
        thread_local std::vector<double> y;
        y.assign((std::size_t)n, 0.0);
        for (Index c = 0; c < n; ++c) {
            const double* column = A + c * ld;
            const double xc = x[c];
            double dot = 0;
            for (Index r = 0; r < c; ++r) {
                y[r] += column[r] * xc;
                dot += column[r] * x[r];
            }
            y[c] += dot + column[c] * xc;
        }
        double total = 0;
        for (Index r = 0; r < n; ++r) total += x[r] * y[r];
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
    if (n >= 11) {
        // the reference's Eigen branch (ML/LinearAlgebra.cpp:50: dest = x x^T): whole columns, unit stride
        for (Index c = 0; c < n; ++c) {
            double* column = dest + c * ld;
            const double xc = x[c];
            for (Index r = 0; r < n; ++r) column[r] = x[r] * xc;
        }
        return;
    }
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
    if (n >= 14) {
        // the reference's Eigen branch (ML/LinearAlgebra.cpp:71: dest += (a x) x^T, a x materialised first): whole columns, unit stride
        thread_local std::vector<double> ax;
        ax.resize((std::size_t)n);
        for (Index r = 0; r < n; ++r) ax[r] = a * x[r];
        for (Index c = 0; c < n; ++c) {
            double* column = dest + c * ld;
            const double xc = x[c];
            for (Index r = 0; r < n; ++r) column[r] += xc * ax[r];
        }
        return;
    }
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
