#pragma once
/* Minimal column-major dense types standing in for the Eigen types of the reference's signatures
 * (Eigen::MatrixXd, Eigen::VectorXd, Eigen::Ref<const Eigen::MatrixXd>, ...), so that the library and its
 * headers need no Eigen. Where Eigen IS available (<Eigen/Core> on the include path) the *Ref views have ONE-step
 * converting constructors from any Eigen dense expression with direct, column-major-compatible storage -- an
 * Eigen::MatrixXd, a block of columns, an Eigen::Map / Eigen::Ref, or the zero-copy transpose of a row-major N x d block as
 * cppyml/clustering.cpp:27-30 passes it -- so reference-style call sites (`em.fit(data)`, `em.fit(x.transpose())`) compile
 * as they are (C++ applies one user-defined conversion implicitly, never two: a constructor from Eigen::Ref alone would
 * not do). Expressions without direct access (sums, products) must be evaluated first, as with Eigen::Ref<T> (non-const).
 * NOT VERIFIED AGAINST REAL EIGEN in this repository's build environment (Eigen is absent there): the branch is parsed
 * and exercised against a small stand-in (tests/cpp/eigen_shim) only; see INTEGRATION.md. */
#include <cstddef>
#include <stdexcept>
#include <type_traits>
#include <vector>

#if __has_include(<Eigen/Core>) && !defined(MLHIP_NO_EIGEN)
#include <Eigen/Core>
#define MLHIP_HAVE_EIGEN 1
#endif

namespace ml {

using Index = std::ptrdiff_t;

#ifdef MLHIP_HAVE_EIGEN
namespace detail {
/// Checks shared by the converting constructors: the expression must expose its coefficients in memory
/// (DirectAccessBit), be laid out column by column (a vector may be either) and have unit inner stride.
template <class Derived> constexpr bool eigen_direct_v = (Eigen::internal::traits<Derived>::Flags & Eigen::DirectAccessBit) != 0;
template <class Derived> void require_column_major(const Eigen::DenseBase<Derived>& m)
{
    static_assert(eigen_direct_v<Derived>, "ml::*Ref needs an Eigen expression with direct memory access: call .eval() first");
    constexpr bool row_major = (Eigen::internal::traits<Derived>::Flags & Eigen::RowMajorBit) != 0;
    if ((row_major && m.rows() != 1 && m.cols() != 1) || m.derived().innerStride() != 1)
        throw std::invalid_argument("ml::*Ref: the Eigen expression is not column-major with unit inner stride");
}
}  // namespace detail
#endif

/** Column-major dynamic matrix (owning). */
class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(Index rows, Index cols) : rows_(rows), cols_(cols), a_(static_cast<std::size_t>(rows * cols)) {}
    void resize(Index rows, Index cols) { rows_ = rows; cols_ = cols; a_.resize(static_cast<std::size_t>(rows * cols)); }
    void setZero() { for (double& v : a_) v = 0.0; }
    void setZero(Index rows, Index cols) { resize(rows, cols); setZero(); }
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index size() const { return rows_ * cols_; }
    Index outerStride() const { return rows_; }
    double* data() { return a_.data(); }
    const double* data() const { return a_.data(); }
    double& operator()(Index i, Index j) { return a_[static_cast<std::size_t>(j * rows_ + i)]; }
    double operator()(Index i, Index j) const { return a_[static_cast<std::size_t>(j * rows_ + i)]; }
    /** Pointer to column j (contiguous, rows() doubles). */
    double* col(Index j) { return a_.data() + j * rows_; }
    const double* col(Index j) const { return a_.data() + j * rows_; }
    void swap(MatrixXd& o) { std::swap(rows_, o.rows_); std::swap(cols_, o.cols_); a_.swap(o.a_); }
#ifdef MLHIP_HAVE_EIGEN
    operator Eigen::Map<const Eigen::MatrixXd>() const { return Eigen::Map<const Eigen::MatrixXd>(data(), rows_, cols_); }
#endif
private:
    Index rows_ = 0, cols_ = 0;
    std::vector<double> a_;
};

/** Dynamic column vector (owning). */
class VectorXd {
public:
    VectorXd() = default;
    explicit VectorXd(Index n) : a_(static_cast<std::size_t>(n)) {}
#ifdef MLHIP_HAVE_EIGEN
    operator Eigen::Map<const Eigen::VectorXd>() const { return Eigen::Map<const Eigen::VectorXd>(data(), size()); }
#endif
    void resize(Index n) { a_.resize(static_cast<std::size_t>(n)); }
    void fill(double v) { for (double& x : a_) x = v; }
    void setZero() { fill(0.0); }
    Index size() const { return static_cast<Index>(a_.size()); }
    double* data() { return a_.data(); }
    const double* data() const { return a_.data(); }
    double& operator[](Index i) { return a_[static_cast<std::size_t>(i)]; }
    double operator[](Index i) const { return a_[static_cast<std::size_t>(i)]; }
    double& operator()(Index i) { return a_[static_cast<std::size_t>(i)]; }
    double operator()(Index i) const { return a_[static_cast<std::size_t>(i)]; }
private:
    std::vector<double> a_;
};

/** Borrowed read-only view of a column-major matrix: the role of Eigen::Ref<const Eigen::MatrixXd>. */
class ConstMatrixRef {
public:
    ConstMatrixRef(const double* p, Index rows, Index cols, Index outer_stride) : p_(p), rows_(rows), cols_(cols), ld_(outer_stride) {}
    ConstMatrixRef(const double* p, Index rows, Index cols) : ConstMatrixRef(p, rows, cols, rows) {}
    ConstMatrixRef(const MatrixXd& m) : ConstMatrixRef(m.data(), m.rows(), m.cols(), m.rows()) {}
#ifdef MLHIP_HAVE_EIGEN
    /** From an Eigen::MatrixXd, a column block, a Map / Ref, or `rowmajor.transpose()` -- one implicit conversion. */
    template <class Derived, class = std::enable_if_t<std::is_same<typename Derived::Scalar, double>::value>>
    ConstMatrixRef(const Eigen::DenseBase<Derived>& m)
        : ConstMatrixRef(m.derived().data(), m.rows(), m.cols(), m.cols() == 1 ? m.rows() : m.derived().outerStride())
    {
        detail::require_column_major(m);
    }
#endif
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index outerStride() const { return ld_; }
    const double* data() const { return p_; }
    const double* col(Index j) const { return p_ + j * ld_; }
    double operator()(Index i, Index j) const { return p_[j * ld_ + i]; }
private:
    const double* p_;
    Index rows_, cols_, ld_;
};

/** Borrowed writable view of a column-major matrix: the role of Eigen::Ref<Eigen::MatrixXd>. */
class MatrixRef {
public:
    MatrixRef(double* p, Index rows, Index cols, Index outer_stride) : p_(p), rows_(rows), cols_(cols), ld_(outer_stride) {}
    MatrixRef(MatrixXd& m) : MatrixRef(m.data(), m.rows(), m.cols(), m.rows()) {}
#ifdef MLHIP_HAVE_EIGEN
    /** From a writable Eigen::MatrixXd / block / Map / Ref (lvalue or the temporary a `.leftCols(k)` call returns). */
    template <class Derived, class = std::enable_if_t<std::is_same<typename Derived::Scalar, double>::value>>
    MatrixRef(Eigen::DenseBase<Derived>& m)
        : MatrixRef(m.derived().data(), m.rows(), m.cols(), m.cols() == 1 ? m.rows() : m.derived().outerStride())
    {
        detail::require_column_major(m);
    }
    template <class Derived, class = std::enable_if_t<std::is_same<typename Derived::Scalar, double>::value && !std::is_const<Derived>::value>>
    MatrixRef(Eigen::DenseBase<Derived>&& m) : MatrixRef(static_cast<Eigen::DenseBase<Derived>&>(m)) {}
#endif
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index outerStride() const { return ld_; }
    double* data() const { return p_; }
    double* col(Index j) const { return p_ + j * ld_; }
    double& operator()(Index i, Index j) const { return p_[j * ld_ + i]; }
    void setZero() const { for (Index j = 0; j < cols_; ++j) for (Index i = 0; i < rows_; ++i) p_[j * ld_ + i] = 0.0; }
private:
    double* p_;
    Index rows_, cols_, ld_;
};

/** Borrowed read-only vector view: the role of Eigen::Ref<const Eigen::VectorXd>. */
class ConstVectorRef {
public:
    ConstVectorRef(const double* p, Index n) : p_(p), n_(n) {}
    ConstVectorRef(const VectorXd& v) : ConstVectorRef(v.data(), v.size()) {}
    ConstVectorRef(const std::vector<double>& v) : ConstVectorRef(v.data(), static_cast<Index>(v.size())) {}
#ifdef MLHIP_HAVE_EIGEN
    /** From an Eigen::VectorXd, a matrix column, a segment, a Map / Ref. @throw std::invalid_argument If it is not a
    contiguous vector. */
    template <class Derived, class = std::enable_if_t<std::is_same<typename Derived::Scalar, double>::value>>
    ConstVectorRef(const Eigen::DenseBase<Derived>& v) : ConstVectorRef(v.derived().data(), v.size())
    {
        static_assert(detail::eigen_direct_v<Derived>, "ml::ConstVectorRef needs an Eigen expression with direct memory access");
        if ((v.rows() != 1 && v.cols() != 1) || (v.size() > 1 && v.derived().innerStride() != 1))
            throw std::invalid_argument("ml::ConstVectorRef: not a contiguous vector");
    }
#endif
    Index size() const { return n_; }
    const double* data() const { return p_; }
    double operator[](Index i) const { return p_[i]; }
private:
    const double* p_;
    Index n_;
};

/** Borrowed writable vector view: the role of Eigen::Ref<Eigen::VectorXd>. */
class VectorRef {
public:
    VectorRef(double* p, Index n) : p_(p), n_(n) {}
    VectorRef(VectorXd& v) : VectorRef(v.data(), v.size()) {}
    VectorRef(std::vector<double>& v) : VectorRef(v.data(), static_cast<Index>(v.size())) {}
#ifdef MLHIP_HAVE_EIGEN
    template <class Derived, class = std::enable_if_t<std::is_same<typename Derived::Scalar, double>::value>>
    VectorRef(Eigen::DenseBase<Derived>& v) : VectorRef(v.derived().data(), v.size())
    {
        static_assert(detail::eigen_direct_v<Derived>, "ml::VectorRef needs an Eigen expression with direct memory access");
        if ((v.rows() != 1 && v.cols() != 1) || (v.size() > 1 && v.derived().innerStride() != 1))
            throw std::invalid_argument("ml::VectorRef: not a contiguous vector");
    }
    template <class Derived, class = std::enable_if_t<std::is_same<typename Derived::Scalar, double>::value && !std::is_const<Derived>::value>>
    VectorRef(Eigen::DenseBase<Derived>&& v) : VectorRef(static_cast<Eigen::DenseBase<Derived>&>(v)) {}
#endif
    Index size() const { return n_; }
    double* data() const { return p_; }
    double& operator[](Index i) const { return p_[i]; }
private:
    double* p_;
    Index n_;
};

}  // namespace ml
