// Parses and runs the `#ifdef MLHIP_HAVE_EIGEN` branches of include/ML/*.hpp. Built against tests/cpp/eigen_shim (a stand-in,
// NOT Eigen -- the real library is absent here), so this checks the adapters' own logic and that reference-style call sites
// are well-formed C++ against them: one-step conversions from a matrix / a block of columns / the transpose of a row-major
// block, the reference-signature initialiser bases, the Eigen-typed LinearAlgebra and calculate_XXt_beta overloads.
// Mode "host" needs no GPU (exact fits, helpers); mode "gpu" adds real fits.
#include <Eigen/Core>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>

#include "ML/Clustering.hpp"
#include "ML/EM.hpp"
#include "ML/KMeans.hpp"
#include "ML/LinearAlgebra.hpp"
#include "ML/LinearRegression.hpp"

#ifndef MLHIP_HAVE_EIGEN
#error "the Eigen branch of Dense.hpp was not selected"
#endif

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

using MatrixXdR = Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;

// written exactly as against the reference (ML/Clustering.hpp:71), only the base class name differs
struct FirstColumns : ml::Clustering::EigenCentroidsInitialiser {
    void init(Eigen::Ref<const Eigen::MatrixXd> data, std::default_random_engine&, unsigned int number_components,
              Eigen::Ref<Eigen::MatrixXd> centroids) const override
    {
        for (unsigned k = 0; k < number_components; ++k)
            for (Eigen::Index j = 0; j < data.rows(); ++j) centroids(j, k) = data(j, k);
    }
};
struct HardFirst : ml::Clustering::EigenResponsibilitiesInitialiser {
    void init(Eigen::Ref<const Eigen::MatrixXd> data, std::default_random_engine&, unsigned int number_components,
              Eigen::Ref<Eigen::MatrixXd> r) const override
    {
        for (Eigen::Index i = 0; i < data.cols(); ++i)
            for (unsigned k = 0; k < number_components; ++k) r(i, k) = (static_cast<unsigned>(i) % number_components == k) ? 1.0 : 0.0;
    }
};

static void host_checks()
{
    // exact fit (N == K) straight from Eigen objects: a MatrixXd, a block of its columns, the transpose of a row-major block
    Eigen::MatrixXd data(3, 4);
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 3; ++i) data(i, j) = 10 * j + i;
    ml::EM em(4);
    CHECK(em.fit(data));                                         // Eigen::MatrixXd -> ml::ConstMatrixRef in ONE conversion
    CHECK(em.means()(2, 3) == 32.0);
    ml::EM em2(2);
    CHECK(em2.fit(data.leftCols(2)));                            // a block of columns (outer stride 3)
    CHECK(em2.means()(1, 1) == 11.0);
    MatrixXdR rows(4, 3);                                        // numpy-style N x d, C-contiguous
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) rows(i, j) = data(j, i);
    ml::Clustering::KMeans km(4);
    CHECK(km.fit(rows.transpose()));                             // cppyml/clustering.cpp:27-30: fit(data.transpose())
    CHECK(km.centroids()(2, 3) == 32.0 && km.inertia() == 0.0);
    bool threw = false;
    try { em.fit(rows); } catch (const std::invalid_argument&) { threw = true; }   // a row-major block itself is refused
    CHECK(threw);

    // the reference-signature initialiser bases, reached through the library's own virtuals
    std::default_random_engine prng;
    ml::MatrixXd c(3, 2);
    FirstColumns fc;
    const ml::Clustering::CentroidsInitialiser& as_base = fc;
    as_base.init(ml::ConstMatrixRef(data), prng, 2, c);
    CHECK(c(1, 1) == 11.0 && c(2, 0) == 2.0);
    ml::MatrixXd r(4, 2);
    HardFirst hf;
    const ml::Clustering::ResponsibilitiesInitialiser& rbase = hf;
    rbase.init(ml::ConstMatrixRef(data), prng, 2, r);
    CHECK(r(0, 0) == 1.0 && r(1, 1) == 1.0 && r(2, 1) == 0.0);

    // LinearAlgebra with the reference's Eigen signatures
    Eigen::MatrixXd A(3, 3);
    Eigen::VectorXd x(3);
    for (int i = 0; i < 3; ++i) { x[i] = i + 1; for (int j = 0; j < 3; ++j) A(i, j) = 1.0 / (1 + i + j); }
    double expected = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) expected += x[i] * A(i, j) * x[j];
    CHECK(std::abs(ml::LinearAlgebra::xAx_symmetric(A, x) - expected) < 1e-14);
    Eigen::MatrixXd xx;
    ml::LinearAlgebra::xxT(x, xx);
    CHECK(xx.rows() == 3 && xx(2, 1) == 6.0);
    ml::LinearAlgebra::add_a_xxT(x, xx, 0.5);
    CHECK(xx(2, 1) == 9.0);
    // point query with Eigen vectors (reference ML/EM.hpp:158)
    Eigen::VectorXd u(4);
    try {
        em.assign_responsibilities(data.col(1), u);              // the call must be well-formed; an exact fit has no
    } catch (const std::invalid_argument&) {                     // covariance decompositions to evaluate it with
    }
}

static void gpu_checks()
{
    std::default_random_engine rng(3);
    std::normal_distribution<double> nrm;
    const int d = 3, n = 600, K = 2;
    MatrixXdR rows(n, d);
    for (int i = 0; i < n; ++i) for (int j = 0; j < d; ++j) rows(i, j) = nrm(rng) * 0.3 + (i % 2 ? 4.0 : -4.0) * (j + 1);
    ml::EM em(K);
    em.set_means_initialiser(std::make_shared<FirstColumns>());
    em.set_responsibilities_initialiser(std::make_shared<HardFirst>());
    em.set_maximise_first(true);                                 // the user-defined ResponsibilitiesInitialiser branch
    CHECK(em.fit(rows.transpose()));
    CHECK(std::abs(em.mixing_probabilities()[0] - 0.5) < 1e-12);
    ml::EM em_b(K);
    em_b.set_means_initialiser(std::make_shared<FirstColumns>());
    CHECK(em_b.fit(rows.transpose()));
    CHECK(std::abs(em_b.log_likelihood() - em.log_likelihood()) < 1e-8 * std::abs(em.log_likelihood()));

    // calculate_XXt_beta with the reference's exact signature (reference ML/LinearRegression.hpp:412)
    const int q = 4, N = 500;
    Eigen::MatrixXd X(q, N);
    Eigen::VectorXd y(N), lambda(q), beta_true(q);
    for (int j = 0; j < q; ++j) { beta_true[j] = j - 1.5; lambda[j] = 0; }
    for (int i = 0; i < N; ++i) {
        double t = 0;
        for (int j = 0; j < q; ++j) { X(j, i) = nrm(rng); t += beta_true[j] * X(j, i); }
        y[i] = t;
    }
    Eigen::MatrixXd XXt(q, q);
    Eigen::LDLT<Eigen::MatrixXd> decomposition;
    const Eigen::VectorXd beta = ml::LinearRegression::calculate_XXt_beta(X, y, XXt, decomposition, lambda);
    for (int j = 0; j < q; ++j) CHECK(std::abs(beta[j] - beta_true[j]) < 1e-10);
    const Eigen::VectorXd again = decomposition.solve(XXt.col(0));   // the decomposition is the caller's to reuse
    CHECK(std::abs(again[0] - 1.0) < 1e-10 && std::abs(again[1]) < 1e-10);
}

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    try {
        host_checks();
        if (gpu) gpu_checks();
    } catch (const std::exception& e) {
        std::printf("FAIL unexpected exception: %s\n", e.what());
        ++failures;
    }
    std::printf(failures ? "%d FAILURES\n" : "OK (%d failures)\n", failures);
    return failures ? 1 : 0;
}
