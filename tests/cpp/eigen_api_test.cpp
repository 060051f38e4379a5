// The Eigen-typed, header-only API (include/ML/EigenApi.hpp, reached through include/eigen_api as the reference's own
// `#include "ML/EM.hpp"` / `"ML/KMeans.hpp"`): the accessor expressions of the reference's tests -- Tests/test_EM.cpp:48-101,
// Tests/test_KMeans.cpp:50-90 -- in the shape they have there, on data drawn by the same libstdc++ calls with the same
// thresholds. Built against tests/cpp/eigen_shim (a stand-in, NOT Eigen: the real library is absent from this environment),
// so this shows that those call sites are well-formed against the header and that the numbers behind the accessors are right;
// it proves nothing about real Eigen. Mode "host": exceptions and the exact fit; mode "gpu": the fits.
#include <Eigen/Core>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "ML/EM.hpp"        // -I include/eigen_api comes first: this is include/ML/EigenApi.hpp
#include "ML/KMeans.hpp"

#ifndef MLHIP_ML_EIGEN_API_HPP
#error "include/eigen_api must precede include/ on the include path"
#endif

static int failures = 0;
#define ASSERT_TRUE(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
#define ASSERT_EQ(a, b) ASSERT_TRUE((a) == (b))
#define ASSERT_LE(a, b) ASSERT_TRUE((a) <= (b))
#define ASSERT_NEAR(a, b, tol) ASSERT_TRUE(std::abs((a) - (b)) <= (tol))
#define ASSERT_THROW(expr, type) do { bool ok_ = false; try { expr; } catch (const type&) { ok_ = true; } catch (...) {} \
    if (!ok_) { std::printf("FAIL %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #type); ++failures; } } while (0)

static void host_checks()
{
    ASSERT_THROW(ml::EM(0), std::invalid_argument);
    ASSERT_THROW(ml::Clustering::KMeans(0), std::invalid_argument);
    ASSERT_THROW(ml::Clustering::ClosestCentroid(nullptr), std::invalid_argument);
    ml::EM em(2);
    ASSERT_THROW(em.set_absolute_tolerance(-1), std::domain_error);
    ASSERT_THROW(em.set_maximum_steps(1), std::invalid_argument);
    ASSERT_THROW(em.set_means_initialiser(nullptr), std::invalid_argument);
    // Tests/test_EM.cpp:126-144 -- as many components as samples: deterministic, no device involved
    Eigen::MatrixXd data(3, 2);
    const double v[6] = {-1, 1, 0.5, 0, 0.5, 0.5};
    for (int i = 0; i < 6; ++i) data.data()[i] = v[i];
    ASSERT_TRUE(em.fit(data));
    ASSERT_NEAR(0., (data - em.means()).norm(), 1e-15);
    ASSERT_EQ(0u, em.labels()[0]);
    ASSERT_EQ(1u, em.labels()[1]);
    ASSERT_THROW(em.covariance(2), std::invalid_argument);
    ml::Clustering::KMeans km(2);
    ASSERT_TRUE(km.fit(data));
    ASSERT_NEAR(0., (data - km.centroids()).norm(), 1e-15);
    ASSERT_EQ(0., km.inertia());
    ml::Clustering::Model& model = km;
    ASSERT_EQ(2u, model.number_clusters());
    Eigen::MatrixXd too_few(3, 1);
    ASSERT_THROW(em.fit(too_few), std::invalid_argument);
}

static void gpu_checks()
{
    // ---- Tests/test_EM.cpp:8-104 (test_em), means initialiser KPP
    std::default_random_engine rng;
    std::uniform_real_distribution<double> u01(0, 1);
    std::normal_distribution<double> n01;
    const unsigned int num_dimensions = 3;
    const unsigned int num_components = 2;
    Eigen::MatrixXd means(num_dimensions, num_components);
    const double mv[6] = {0.4, 0.11, 0.5, -1.2, 2.2, 1.6};
    for (int i = 0; i < 6; ++i) means.data()[i] = mv[i];
    Eigen::MatrixXd sigmas(num_dimensions, num_components);
    const double sv[6] = {0.05, 0.04, 0.01, 0.2, 0.1, 0.2};
    for (int i = 0; i < 6; ++i) sigmas.data()[i] = sv[i];
    constexpr double p0 = 0.25;
    const unsigned int sample_size = 400;
    Eigen::MatrixXd data(num_dimensions, sample_size);
    std::vector<unsigned int> ground_truth_labels(sample_size);
    for (unsigned int i = 0; i < sample_size; ++i) {
        const unsigned int k = u01(rng) < p0 ? 0 : 1;
        ground_truth_labels[i] = k;
        for (unsigned int l = 0; l < num_dimensions; ++l) data(l, i) = n01(rng) * sigmas(l, k) + means(l, k);
    }
    {
        ml::EM em(num_components);
        em.set_absolute_tolerance(1e-8);
        em.set_relative_tolerance(1e-8);
        em.set_maximum_steps(100);
        em.set_means_initialiser(std::make_shared<ml::Clustering::KPP>());
        em.set_maximise_first(false);
        const unsigned int seed = 63413131;
        em.set_seed(seed);
        ASSERT_TRUE(em.fit(data));
        ASSERT_TRUE(em.converged());
        ASSERT_EQ(num_components, static_cast<unsigned int>(em.mixing_probabilities().size()));
        ASSERT_EQ(sample_size, static_cast<unsigned int>(em.labels().size()));
        ASSERT_EQ(num_components, static_cast<unsigned int>(em.means().cols()));
        ASSERT_EQ(num_dimensions, static_cast<unsigned int>(em.means().rows()));
        ASSERT_EQ(sample_size, static_cast<unsigned int>(em.responsibilities().rows()));
        ASSERT_EQ(num_components, static_cast<unsigned int>(em.responsibilities().cols()));
        const Eigen::MatrixXd means_col_major(em.means());
        ASSERT_EQ(num_dimensions, static_cast<unsigned int>(means_col_major.rows()));

        Eigen::VectorXd u(num_components);
        for (unsigned int i = 0; i < sample_size; ++i) {
            em.assign_responsibilities(data.col(i), u);
            ASSERT_NEAR(0, (u - em.responsibilities().row(i).transpose()).norm(), 1e-15);
        }

        std::vector<Eigen::MatrixXd> covariances(num_components);
        for (unsigned int k = 0; k < num_components; ++k) {
            const auto sigma_vec = sigmas.col(k);
            covariances[k].setZero(num_dimensions, num_dimensions);
            for (unsigned int l = 0; l < num_dimensions; ++l) covariances[k](l, l) = std::pow(sigma_vec[l], 2);
        }
        Eigen::VectorXd mixing_probabilities(num_components);
        mixing_probabilities[0] = p0;
        mixing_probabilities[1] = 1 - p0;
        // EM could have discovered the clusters in either order.
        constexpr bool first_p_lower = p0 < 1 - p0;
        if ((em.mixing_probabilities()[0] < em.mixing_probabilities()[1]) != first_p_lower) {
            std::swap(mixing_probabilities[0], mixing_probabilities[1]);
            means.col(0).swap(means.col(1));
            std::swap(covariances[0], covariances[1]);
        }
        ASSERT_NEAR(0., (mixing_probabilities - em.mixing_probabilities()).norm(), 2e-2);
        ASSERT_NEAR(0., (means - em.means()).norm(), 2e-2);
        for (unsigned int k = 0; k < num_components; ++k) ASSERT_NEAR(0., (covariances[k] - em.covariance(k)).norm(), 1e-2);
        ASSERT_EQ(num_components, static_cast<unsigned int>(em.covariances().size()));

        ml::EM em1(1);
        em1.set_means_initialiser(std::make_shared<ml::Clustering::KPP>());
        em1.fit(data);
        ASSERT_LE(em1.log_likelihood(), em.log_likelihood());
        ASSERT_NEAR(0., (data.rowwise().mean() - em1.means().col(0)).norm(), 1e-14);
        u.resize(1);
        for (unsigned int i = 0; i < sample_size; ++i) {
            em1.assign_responsibilities(data.col(i), u);
            ASSERT_NEAR(0, (u - em1.responsibilities().row(i).transpose()).norm(), 1e-15);
            ASSERT_EQ(0u, em1.labels()[i]);
        }
        // the other library initialisers through the same surface (Tests/test_EM.cpp:106-124)
        ml::EM em_cc(num_components);
        em_cc.set_responsibilities_initialiser(std::make_shared<ml::Clustering::ClosestCentroid>(std::make_shared<ml::Clustering::Forgy>()));
        em_cc.set_maximise_first(true);
        em_cc.set_seed(seed);
        em_cc.set_maximum_steps(100);
        ASSERT_TRUE(em_cc.fit(data));
        ASSERT_NEAR(em_cc.log_likelihood(), em.log_likelihood(), 1e-6 * std::abs(em.log_likelihood()));
    }
    // ---- Tests/test_KMeans.cpp:8-91 (test_kmeans), default initialiser (`means` may have been swapped above: start from the
    // generating values again)
    {
        Eigen::MatrixXd centroids(num_dimensions, num_components);
        for (int i = 0; i < 6; ++i) centroids.data()[i] = mv[i];
        const unsigned int num_clusters = 2;
        ml::Clustering::KMeans km(num_clusters);
        km.set_absolute_tolerance(1e-8);
        km.set_maximum_steps(100);
        const unsigned int seed = 63413131;
        km.set_seed(seed);
        ASSERT_TRUE(km.fit(data));
        ASSERT_EQ(num_clusters, static_cast<unsigned int>(km.centroids().cols()));
        ASSERT_EQ(num_dimensions, static_cast<unsigned int>(km.centroids().rows()));
        ASSERT_EQ(sample_size, static_cast<unsigned int>(km.labels().size()));
        const Eigen::MatrixXd centroids_col_major(km.centroids());
        ASSERT_EQ(num_clusters, static_cast<unsigned int>(centroids_col_major.cols()));

        double inertia = 0;
        for (unsigned int i = 0; i < sample_size; ++i) {
            const auto label_and_distance = km.assign_label(data.col(i));
            ASSERT_EQ(label_and_distance.first, km.labels()[i]);
            ASSERT_NEAR((km.centroids().col(label_and_distance.first) - data.col(i)).squaredNorm(), label_and_distance.second, 1e-15);
            inertia += label_and_distance.second;
        }
        ASSERT_NEAR(inertia, km.inertia(), 1e-15);
        // KMeans could have discovered the clusters in either order.
        if (ground_truth_labels[0] != km.labels()[0]) {
            for (unsigned int i = 0; i < sample_size; ++i) ground_truth_labels[i] = 1 - ground_truth_labels[i];
            centroids.col(0).swap(centroids.col(1));
        }
        ASSERT_NEAR(0., (centroids - km.centroids()).norm(), 2e-2);
        ASSERT_TRUE(ground_truth_labels == km.labels());

        km.set_seed(seed);
        km.set_number_initialisations(3);
        ASSERT_TRUE(km.fit(data));
        ASSERT_LE(km.inertia(), inertia);

        ml::Clustering::KMeans km1(1);
        km1.fit(data);
        ASSERT_NEAR(0., (data.rowwise().mean() - km1.centroids().col(0)).norm(), 1e-14);
        for (unsigned int i = 0; i < sample_size; ++i) ASSERT_EQ(0u, km1.assign_label(data.col(i)).first);
    }
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
