// C++ drop-in check of the facade headers (include/ML/*.hpp): compiled with plain g++ (no hipcc, no Eigen) against
// libmlhip.so. Mirrors the structure of the reference's Tests/test_EM.cpp / test_KMeans.cpp / test_LinearAlgebra.cpp:
// mode "host" runs what needs no GPU (exceptions, exact fits, helpers), mode "gpu" adds full fits.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <stdexcept>
#include <vector>

#include "ML/Clustering.hpp"
#include "ML/EM.hpp"
#include "ML/KMeans.hpp"
#include "ML/LinearAlgebra.hpp"
#include "ML/LinearRegression.hpp"
#include "ML/Device.hpp"
#include "mlhip.h"
#include <unistd.h>
#include <string>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
#define CHECK_THROWS(expr, type) do { bool ok_ = false; try { expr; } catch (const type&) { ok_ = true; } catch (...) {} \
    if (!ok_) { std::printf("FAIL %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #type); ++failures; } } while (0)

using ml::MatrixXd;
using ml::VectorXd;

static void host_checks()
{
    // constructor / setter errors (ML/EM.cpp:35-82, ML/KMeans.cpp:21,124-148)
    CHECK_THROWS(ml::EM(0), std::invalid_argument);
    ml::EM em(2);
    CHECK(!em.converged());
    CHECK(em.number_components() == 2 && em.number_clusters() == 2);
    CHECK_THROWS(em.set_absolute_tolerance(-1), std::domain_error);
    CHECK_THROWS(em.set_relative_tolerance(-1), std::domain_error);
    CHECK_THROWS(em.set_maximum_steps(1), std::invalid_argument);
    CHECK_THROWS(em.set_means_initialiser(nullptr), std::invalid_argument);
    CHECK_THROWS(em.set_responsibilities_initialiser(nullptr), std::invalid_argument);
    CHECK_THROWS(em.covariance(2), std::invalid_argument);
    CHECK_THROWS(ml::Clustering::ClosestCentroid(nullptr), std::invalid_argument);
    CHECK_THROWS(ml::Clustering::KMeans(0), std::invalid_argument);
    ml::Clustering::KMeans km(2);
    CHECK_THROWS(km.set_absolute_tolerance(-1), std::domain_error);
    CHECK_THROWS(km.set_maximum_steps(1), std::invalid_argument);
    CHECK_THROWS(km.set_number_initialisations(0), std::invalid_argument);
    CHECK_THROWS(km.set_centroids_initialiser(nullptr), std::invalid_argument);

    // deterministic exact fits (Tests/test_EM.cpp:126-144, Tests/test_KMeans.cpp:108-128); data 3 x 2, column = sample
    MatrixXd data(3, 2);
    const double v[6] = {-1, 1, 0.5, 0, 0.5, 0.5};
    std::memcpy(data.data(), v, sizeof(v));
    CHECK(em.fit(data));
    CHECK(km.fit(data));
    CHECK(km.inertia() == 0.0);
    for (unsigned i = 0; i < 2; ++i) {
        CHECK(em.labels()[i] == i && km.labels()[i] == i);
        for (int j = 0; j < 3; ++j) CHECK(em.means()(j, i) == data(j, i) && km.centroids()(j, i) == data(j, i));
    }
    ml::Clustering::Model& as_model = em;   // the abstract interface is intact
    CHECK(as_model.converged() && as_model.centroids().cols() == 2);
    MatrixXd too_few(3, 1);
    CHECK_THROWS(em.fit(too_few), std::invalid_argument);
    MatrixXd no_rows(0, 5);
    CHECK_THROWS(em.fit(no_rows), std::invalid_argument);
    CHECK_THROWS(km.fit(too_few), std::invalid_argument);

    // LinearAlgebra (Tests/test_LinearAlgebra.cpp)
    std::default_random_engine rng(5);
    std::uniform_real_distribution<double> u(-1, 1);
    for (int n : {4, 1024}) {
        MatrixXd A(n, n);
        VectorXd x(n);
        for (int i = 0; i < n; ++i) { x[i] = u(rng); for (int j = 0; j <= i; ++j) A(i, j) = A(j, i) = u(rng); }
        double expected = 0;
        for (int i = 0; i < n; ++i) { double t = 0; for (int j = 0; j < n; ++j) t += A(i, j) * x[j]; expected += x[i] * t; }
        CHECK(std::abs(ml::LinearAlgebra::xAx_symmetric(A, x) - expected) <= std::abs(expected) * 1e-14);
        MatrixXd xx;
        ml::LinearAlgebra::xxT(x, xx);
        MatrixXd B(A);
        ml::LinearAlgebra::add_a_xxT(x, B, 0.6);
        double err = 0, norm = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            err += std::pow(xx(i, j) - x[i] * x[j], 2) + std::pow(B(i, j) - (A(i, j) + 0.6 * x[i] * x[j]), 2);
            norm += std::pow(A(i, j) + 0.6 * x[i] * x[j], 2);
        }
        CHECK(std::sqrt(err) <= std::sqrt(norm) * 1e-15);
    }
    MatrixXd A23(2, 3);
    VectorXd x2(2);
    CHECK_THROWS(ml::LinearAlgebra::xAx_symmetric(A23, x2), std::invalid_argument);
    MatrixXd A33(3, 3);
    CHECK_THROWS(ml::LinearAlgebra::xAx_symmetric(A33, x2), std::invalid_argument);
    CHECK_THROWS(ml::LinearAlgebra::add_a_xxT(x2, A33, 0.6), std::invalid_argument);
}

// User-defined extension points (reference ML/Clustering.hpp:58-89): subclasses written by the library's user.
struct EveryOtherSample : ml::Clustering::CentroidsInitialiser {
    void init(ml::ConstMatrixRef data, std::default_random_engine&, unsigned int number_components, ml::MatrixRef centroids) const override
    {
        for (unsigned k = 0; k < number_components; ++k)
            for (ml::Index j = 0; j < data.rows(); ++j) centroids(j, k) = data(j, 2 * k);
    }
};
struct SoftByFirstCoordinate : ml::Clustering::ResponsibilitiesInitialiser {
    mutable unsigned calls = 0;
    void init(ml::ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, ml::MatrixRef r) const override
    {
        ++calls;
        std::uniform_real_distribution<double> u(0.05, 0.15);
        for (ml::Index i = 0; i < data.cols(); ++i) {
            const unsigned home = data(0, i) > -0.4 ? 0u : 1u;         // the two true clusters of gpu_checks()
            const double leak = u(prng);
            for (unsigned k = 0; k < number_components; ++k)
                r(i, k) = k == home ? 1.0 - leak * (number_components - 1) : leak;
        }
    }
};

// ml::LDLT (stands in for Eigen::LDLT in calculate_XXt_beta's signature): positive definite, semi-definite and negative cases
static void ldlt_checks()
{
    std::default_random_engine rng(9);
    std::normal_distribution<double> nrm;
    const int n = 7;
    MatrixXd B(n, n), A(n, n);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) B(i, j) = nrm(rng);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) { double t = 0; for (int l = 0; l < n; ++l) t += B(i, l) * B(j, l); A(i, j) = t; }
    ml::LDLT ldlt(A);
    CHECK(ldlt.isPositive() && !ldlt.isNegative());
    const MatrixXd R = ldlt.reconstructedMatrix();
    double err = 0, norm = 0;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) { err += std::pow(R(i, j) - A(i, j), 2); norm += A(i, j) * A(i, j); }
    CHECK(std::sqrt(err) <= 1e-13 * std::sqrt(norm));
    VectorXd x0(n), b(n);
    for (int i = 0; i < n; ++i) x0[i] = i - 2.5;
    for (int i = 0; i < n; ++i) { double t = 0; for (int j = 0; j < n; ++j) t += A(i, j) * x0[j]; b[i] = t; }
    const VectorXd x = ldlt.solve(b);
    for (int i = 0; i < n; ++i) CHECK(std::abs(x[i] - x0[i]) <= 1e-9);
    // the largest diagonal entry is pivoted to the front
    double biggest = 0;
    for (int i = 0; i < n; ++i) biggest = std::max(biggest, A(i, i));
    CHECK(A(ldlt.transpositions()[0], ldlt.transpositions()[0]) == biggest);
    // rank-deficient (two identical features, no ridge): still a solution of the normal equations
    MatrixXd S(3, 3);
    const double sv[9] = {2, 2, 1, 2, 2, 1, 1, 1, 3};
    std::memcpy(S.data(), sv, sizeof(sv));
    ml::LDLT semi(S);
    CHECK(semi.isPositive());
    VectorXd rhs(3);
    rhs[0] = 4; rhs[1] = 4; rhs[2] = 5;                              // = S * (1, 1, 1)
    const VectorXd z = semi.solve(rhs);
    for (int i = 0; i < 3; ++i) { double t = 0; for (int j = 0; j < 3; ++j) t += S(i, j) * z[j]; CHECK(std::abs(t - rhs[i]) <= 1e-12); }
    MatrixXd N(A);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) N(i, j) = -A(i, j);
    CHECK(ml::LDLT(N).isNegative() && !ml::LDLT(N).isPositive());
    // indefinite: neither positive nor negative (Eigen::LDLT::isPositive is true only for a positive or zero sign)
    MatrixXd I2(2, 2);
    I2(0, 0) = 1; I2(1, 1) = -2; I2(0, 1) = I2(1, 0) = 0;
    const ml::LDLT indef(I2);
    CHECK(indef.isIndefinite() && !indef.isPositive() && !indef.isNegative());
    CHECK_THROWS(ml::LDLT(MatrixXd(2, 3)), std::invalid_argument);
    CHECK_THROWS(ldlt.solve(VectorXd(n + 1)), std::invalid_argument);
}

// Tests/test_EM.cpp:8-104 and Tests/test_KMeans.cpp:8-106 (same libstdc++ draws, same invariants)
static void gpu_checks()
{
    std::default_random_engine rng;
    std::uniform_real_distribution<double> u01(0, 1);
    std::normal_distribution<double> standard_normal;
    const unsigned K = 2, d = 3, n = 400;
    const double p0 = 0.25;
    const double means[2][3] = {{0.4, 0.11, 0.5}, {-1.2, 2.2, 1.6}};
    const double sigmas[2][3] = {{0.05, 0.04, 0.01}, {0.2, 0.1, 0.2}};
    MatrixXd data(d, n);
    std::vector<unsigned> truth(n);
    for (unsigned i = 0; i < n; ++i) {
        const unsigned k = u01(rng) < p0 ? 0 : 1;
        truth[i] = k;
        for (unsigned l = 0; l < d; ++l) data(l, i) = standard_normal(rng) * sigmas[k][l] + means[k][l];
    }

    ml::EM em(K);
    em.set_absolute_tolerance(1e-8);
    em.set_relative_tolerance(1e-8);
    em.set_maximum_steps(100);
    em.set_means_initialiser(std::make_shared<ml::Clustering::KPP>());
    em.set_seed(63413131);
    CHECK(em.fit(data));
    CHECK(em.converged());
    CHECK(em.labels().size() == n && em.means().rows() == d && em.means().cols() == K);
    CHECK(em.responsibilities().rows() == n && em.responsibilities().cols() == K);
    VectorXd uu(K);
    for (unsigned i = 0; i < n; ++i) {
        em.assign_responsibilities(ml::ConstVectorRef(data.col(i), d), uu);
        double e2 = 0;
        for (unsigned k = 0; k < K; ++k) e2 += std::pow(uu[k] - em.responsibilities()(i, k), 2);
        CHECK(std::sqrt(e2) <= 1e-15);
    }
    const bool swap = (em.mixing_probabilities()[0] < em.mixing_probabilities()[1]) != (p0 < 1 - p0);
    for (unsigned k = 0; k < K; ++k) {
        const unsigned t = swap ? 1 - k : k;
        CHECK(std::abs(em.mixing_probabilities()[k] - (t == 0 ? p0 : 1 - p0)) <= 2e-2);
        for (unsigned l = 0; l < d; ++l) {
            CHECK(std::abs(em.means()(l, k) - means[t][l]) <= 2e-2);
            CHECK(std::abs(em.covariance(k)(l, l) - sigmas[t][l] * sigmas[t][l]) <= 1e-2);
        }
    }
    ml::EM em1(1);
    em1.fit(data);
    CHECK(em1.log_likelihood() <= em.log_likelihood());

    ml::Clustering::KMeans km(K);
    km.set_absolute_tolerance(1e-8);
    km.set_maximum_steps(100);
    km.set_seed(63413131);
    CHECK(km.fit(data));
    double inertia = 0;
    for (unsigned i = 0; i < n; ++i) {
        const auto ld = km.assign_label(ml::ConstVectorRef(data.col(i), d));
        CHECK(ld.first == km.labels()[i]);
        inertia += ld.second;
    }
    CHECK(std::abs(inertia - km.inertia()) <= 1e-15);
    const bool kswap = truth[0] != km.labels()[0];
    for (unsigned i = 0; i < n; ++i) CHECK((kswap ? 1 - truth[i] : truth[i]) == km.labels()[i]);
    km.set_seed(63413131);
    km.set_number_initialisations(3);
    CHECK(km.fit(data));
    CHECK(km.inertia() <= inertia);

    // user-defined initialisers through both start modes of EM::fit (reference ML/EM.cpp:120-135) and through KMeans
    auto soft = std::make_shared<SoftByFirstCoordinate>();
    ml::EM em_user(K);
    em_user.set_absolute_tolerance(1e-8);
    em_user.set_relative_tolerance(1e-8);
    em_user.set_maximum_steps(100);
    em_user.set_responsibilities_initialiser(soft);
    em_user.set_maximise_first(true);
    CHECK(em_user.fit(data));
    CHECK(soft->calls == 1);
    CHECK(std::abs(em_user.log_likelihood() - em.log_likelihood()) <= 1e-6 * std::abs(em.log_likelihood()));
    for (unsigned i = 0; i < n; ++i) CHECK((em_user.labels()[i] == em_user.labels()[0]) == (truth[i] == truth[0]));
    ml::EM em_means(K);
    em_means.set_absolute_tolerance(1e-8);
    em_means.set_relative_tolerance(1e-8);
    em_means.set_maximum_steps(100);
    em_means.set_means_initialiser(std::make_shared<EveryOtherSample>());
    CHECK(em_means.fit(data));
    CHECK(std::abs(em_means.log_likelihood() - em.log_likelihood()) <= 1e-6 * std::abs(em.log_likelihood()));
    ml::Clustering::KMeans km_user(K);
    km_user.set_centroids_initialiser(std::make_shared<EveryOtherSample>());
    CHECK(km_user.fit(data));
    CHECK(std::abs(km_user.inertia() - inertia) <= 1e-12 * inertia);

    // calculate_XXt_beta in the reference's 5-argument shape (reference ML/LinearRegression.cpp:201-230)
    const unsigned q = 5, N = 2000;
    MatrixXd X(q, N);
    VectorXd y(N), lambda(q), beta_true(q);
    for (unsigned j = 0; j < q; ++j) { beta_true[j] = 0.5 * j - 1.0; lambda[j] = 0.0; }
    for (unsigned i = 0; i < N; ++i) {
        double t = 0;
        for (unsigned j = 0; j < q; ++j) { X(j, i) = standard_normal(rng); t += beta_true[j] * X(j, i); }
        y[i] = t + 1e-3 * standard_normal(rng);
    }
    MatrixXd XXt(q, q);
    ml::LDLT decomposition;
    const VectorXd beta = ml::LinearRegression::calculate_XXt_beta(X, y, XXt, decomposition, lambda);
    for (unsigned j = 0; j < q; ++j) CHECK(std::abs(beta[j] - beta_true[j]) <= 1e-3);
    CHECK(decomposition.rows() == q && decomposition.isPositive());
    VectorXd e0(q);
    e0.setZero();
    e0[0] = 1;
    const VectorXd inv_col = decomposition.solve(e0);               // what ols / ridge reuse it for: (X X^T)^-1 columns
    double t0 = 0;
    for (unsigned j = 0; j < q; ++j) t0 += XXt(0, j) * inv_col[j];
    CHECK(std::abs(t0 - 1.0) <= 1e-10);
    // collinear features, no ridge: the reference's pivoted LDLT still returns a least-squares solution
    for (unsigned i = 0; i < N; ++i) X(1, i) = X(0, i);
    const VectorXd beta2 = ml::LinearRegression::calculate_XXt_beta(X, y, XXt, decomposition, lambda);
    double resid = 0, total = 0;
    for (unsigned i = 0; i < N; ++i) {
        double t = 0;
        for (unsigned j = 0; j < q; ++j) t += beta2[j] * X(j, i);
        resid += std::pow(y[i] - t, 2);
        total += y[i] * y[i];
    }
    CHECK(std::isfinite(resid) && resid < total);
    lambda[2] = -1;
    CHECK_THROWS(ml::LinearRegression::calculate_XXt_beta(X, y, XXt, decomposition, lambda), std::domain_error);

    // Native RCCL from C++ (no Python, no hook): a context with the library's own communicator becomes the facade's context.
    // (One GPU here: a 1-rank communicator -- RCCL refuses two ranks on one device; every step still goes through
    // ncclAllReduce on the context's stream.)
    mlhip_ctx* rccl_ctx = nullptr;
    CHECK(mlhip_ctx_create(0, &rccl_ctx) == MLHIP_OK);
    const std::string id_file = "/tmp/mlhip_rccl_id_" + std::to_string(static_cast<long>(getpid()));
    const int rc = mlhip_ctx_init_rccl_file(rccl_ctx, id_file.c_str(), 1, 0);
    if (rc != MLHIP_OK) std::printf("FAIL mlhip_ctx_init_rccl_file: %s\n", mlhip_last_error());
    CHECK(rc == MLHIP_OK);
    int ranks = 0;
    CHECK(mlhip_ctx_rccl_ranks(rccl_ctx, &ranks) == MLHIP_OK && ranks == 1);
    ml::device::set_context(rccl_ctx);
    ml::EM em_rccl(K);
    em_rccl.set_absolute_tolerance(1e-8);
    em_rccl.set_relative_tolerance(1e-8);
    em_rccl.set_maximum_steps(100);
    em_rccl.set_means_initialiser(std::make_shared<ml::Clustering::KPP>());
    em_rccl.set_seed(63413131);
    CHECK(em_rccl.fit(data));
    CHECK(em_rccl.log_likelihood() == em.log_likelihood());      // a sum over one rank is the identity: bit-identical
    CHECK(em_rccl.steps_done() == em.steps_done());
    em_rccl.release_device_data();                               // (its HBM block belongs to rccl_ctx)
    ml::device::set_context(nullptr);
    CHECK(mlhip_ctx_destroy(rccl_ctx) == MLHIP_OK);
    unlink(id_file.c_str());
}

// The reference's API is ONE process handing ONE d x N block to fit (ML/EM.cpp:91, ML/KMeans.cpp:25). Through a device group the same
// call drives several GPUs (here: several shards on GPU 0 -- the in-process all-reduce): `explicit_group` builds the group through
// the C ABI and installs it, otherwise the facade's own default context must already be one (MLHIP_NUM_GPUS / MLHIP_DEVICES in the
// environment: mode "env-group").
static void group_checks(bool explicit_group)
{
    std::default_random_engine rng(11);
    std::normal_distribution<double> standard_normal;
    const unsigned K = 3, d = 5, n = 30001;
    MatrixXd data(d, n);
    for (unsigned i = 0; i < n; ++i)
        for (unsigned l = 0; l < d; ++l) data(l, i) = standard_normal(rng) + 5.0 * (i % K) * ((l & 1) ? 1.0 : -0.5);   // three well separated clusters
    auto fit_em = [&](ml::EM& em) {
        em.set_absolute_tolerance(1e-9);
        em.set_relative_tolerance(0);
        em.set_maximum_steps(200);
        em.set_means_initialiser(std::make_shared<ml::Clustering::KPP>());
        em.set_seed(4242);
        return em.fit(data);
    };
    ml::EM one(K);
    ml::Clustering::KMeans km_one(K);
    km_one.set_seed(99);
    km_one.set_centroids_initialiser(std::make_shared<ml::Clustering::RandomPartition>());
    mlhip_ctx* single = nullptr;
    CHECK(mlhip_ctx_create(0, &single) == MLHIP_OK);
    ml::device::set_context(single);
    CHECK(fit_em(one));
    CHECK(km_one.fit(data));
    one.release_device_data();
    ml::device::set_context(nullptr);

    mlhip_ctx* group = nullptr;
    if (explicit_group) {
        const int devices[4] = {0, 0, 0, 0};
        CHECK(mlhip_ctx_create_group(4, devices, &group) == MLHIP_OK);
        ml::device::set_context(group);
    }
    int shards = 0, world = 0, rank = -1;
    const char* kind = "";
    CHECK(mlhip_ctx_shards(ml::device::context(), &shards) == MLHIP_OK && shards == 4);
    CHECK(mlhip_ctx_world(ml::device::context(), &world, &rank) == MLHIP_OK && world == 1 && rank == 0);
    CHECK(mlhip_ctx_reduce_kind(ml::device::context(), &kind) == MLHIP_OK && std::strncmp(kind, "group-", 6) == 0);
    ml::EM many(K);
    CHECK(fit_em(many));
    CHECK(many.steps_done() == one.steps_done());
    CHECK(std::abs(many.log_likelihood() - one.log_likelihood()) <= 1e-12 * std::abs(one.log_likelihood()));
    CHECK(many.labels() == one.labels());
    for (unsigned k = 0; k < K; ++k) {
        CHECK(std::abs(many.mixing_probabilities()[k] - one.mixing_probabilities()[k]) <= 1e-11);
        for (unsigned l = 0; l < d; ++l) CHECK(std::abs(many.means()(l, k) - one.means()(l, k)) <= 1e-10);
    }
    CHECK(many.responsibilities().rows() == n);
    double worst = 0;
    for (unsigned i = 0; i < n; i += 97)
        for (unsigned k = 0; k < K; ++k) worst = std::max(worst, std::abs(many.responsibilities()(i, k) - one.responsibilities()(i, k)));
    CHECK(worst <= 1e-11);
    ml::Clustering::KMeans km_many(K);
    km_many.set_seed(99);
    km_many.set_centroids_initialiser(std::make_shared<ml::Clustering::RandomPartition>());
    CHECK(km_many.fit(data));
    CHECK(km_many.labels() == km_one.labels());
    CHECK(std::abs(km_many.inertia() - km_one.inertia()) <= 1e-12 * km_one.inertia());
    many.release_device_data();
    if (explicit_group) {
        ml::device::set_context(nullptr);
        CHECK(mlhip_ctx_destroy(group) == MLHIP_OK);
    }
    CHECK(mlhip_ctx_destroy(single) == MLHIP_OK);
}

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    const bool env_group = argc > 1 && std::strcmp(argv[1], "env-group") == 0;
    try {
        if (env_group) {
            group_checks(false);
            std::printf(failures ? "%d FAILURES\n" : "OK (%d failures)\n", failures);
            return failures ? 1 : 0;
        }
        host_checks();
        ldlt_checks();
        if (gpu) gpu_checks();
        if (gpu) group_checks(true);
    } catch (const std::exception& e) {
        std::printf("FAIL unexpected exception: %s\n", e.what());
        ++failures;
    }
    std::printf(failures ? "%d FAILURES\n" : "OK (%d failures)\n", failures);
    return failures ? 1 : 0;
}
