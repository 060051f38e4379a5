#pragma once
/* The reference's class API with the reference's OWN types: header-only ml::EM, ml::Clustering::KMeans and the library
 * initialisers whose accessors ARE Eigen objects -- `const Eigen::MatrixXd& means()`, `const std::vector<Eigen::MatrixXd>&
 * covariances()`, `const Eigen::VectorXd& mixing_probabilities()`, `const Eigen::MatrixXd& responsibilities()`,
 * `const Eigen::MatrixXd& centroids()` (reference ML/EM.hpp:100-135, ML/KMeans.hpp:34-60) -- so that code which chains Eigen on
 * them, like the reference's own tests (`(means - em.means()).norm()`, `em.responsibilities().row(i).transpose()`,
 * Tests/test_EM.cpp:56-101, Tests/test_KMeans.cpp:57-90), compiles as written. The classes are thin owners of the flat C handles
 * of include/mlpp_c.h (the same handles the Python surface uses): every call goes to libmlhip.so, results are copied into Eigen
 * members after fit() -- responsibilities lazily, on first access, like the C++ facade of include/ML/EM.hpp.
 *
 * Use INSTEAD of include/ML/{EM,KMeans,Clustering}.hpp in a translation unit (the two families share their names; the classes
 * here live in the inline namespace ml::eigen_api so that their symbols never meet the library's): put include/eigen_api in
 * front of include/ on the include path and `#include "ML/EM.hpp"` resolves to this header.
 * Limits: user-defined initialisers cannot cross the C handles -- subclass the initialiser bases of include/ML/Clustering.hpp
 * (EigenCentroidsInitialiser) with the C++ facade for that; LinearAlgebra / LinearRegression helpers are in their own headers.
 * NOT VERIFIED AGAINST REAL EIGEN in this repository's build environment (Eigen is absent there): compiled and run against the
 * stand-in tests/cpp/eigen_shim only (tests/cpp/eigen_api_test.cpp); see INTEGRATION.md. */
#if defined(MLHIP_ML_EM_HPP) || defined(MLHIP_ML_KMEANS_HPP) || defined(MLHIP_ML_CLUSTERING_HPP)
#error "ML/EigenApi.hpp replaces ML/EM.hpp / ML/KMeans.hpp / ML/Clustering.hpp in a translation unit: include one family only"
#endif
#define MLHIP_ML_EIGEN_API_HPP

#include <Eigen/Core>

#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../mlhip.h"
#include "../mlpp_c.h"

namespace ml {
inline namespace eigen_api {

namespace detail {
/// A failed C call as the exception the reference throws there (ML/EM.cpp:35-100, ML/KMeans.cpp:21-148).
inline void check(int status)
{
    if (status == MLHIP_OK) return;
    const std::string msg = mlhip_last_error();
    if (status == MLHIP_E_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (status == MLHIP_E_DOMAIN) throw std::domain_error(msg);
    throw std::runtime_error(msg);
}
/// The d x N block as the contiguous memory the C handles take (a copy only when the columns are strided).
struct Block {
    Eigen::MatrixXd copy;
    const double* p;
    explicit Block(Eigen::Ref<const Eigen::MatrixXd> data) : p(data.data())
    {
        if (data.outerStride() != data.rows()) { copy = data; p = copy.data(); }
    }
};
}  // namespace detail

namespace Clustering {

/** Initialisers (ML/Clustering.hpp:58-125): owners of the library's implementations. */
class CentroidsInitialiser {
public:
    virtual ~CentroidsInitialiser() { if (h_) mlpp_centroids_initialiser_destroy(h_); }
    CentroidsInitialiser(const CentroidsInitialiser&) = delete;
    CentroidsInitialiser& operator=(const CentroidsInitialiser&) = delete;
    const mlpp_centroids_initialiser* handle() const { return h_; }
protected:
    CentroidsInitialiser() = default;
    mlpp_centroids_initialiser* h_ = nullptr;
};
class Forgy : public CentroidsInitialiser { public: Forgy() { detail::check(mlpp_forgy_create(&h_)); } };
class RandomPartition : public CentroidsInitialiser { public: RandomPartition() { detail::check(mlpp_random_partition_create(&h_)); } };
class KPP : public CentroidsInitialiser { public: KPP() { detail::check(mlpp_kpp_create(&h_)); } };

class ResponsibilitiesInitialiser {
public:
    virtual ~ResponsibilitiesInitialiser() { if (h_) mlpp_responsibilities_initialiser_destroy(h_); }
    ResponsibilitiesInitialiser(const ResponsibilitiesInitialiser&) = delete;
    ResponsibilitiesInitialiser& operator=(const ResponsibilitiesInitialiser&) = delete;
    const mlpp_responsibilities_initialiser* handle() const { return h_; }
protected:
    ResponsibilitiesInitialiser() = default;
    mlpp_responsibilities_initialiser* h_ = nullptr;
};
class ClosestCentroid : public ResponsibilitiesInitialiser {
public:
    /** @throw std::invalid_argument If `centroids_initialiser` is null (ML/Clustering.cpp:68). */
    explicit ClosestCentroid(std::shared_ptr<const CentroidsInitialiser> centroids_initialiser) : centroids_initialiser_(centroids_initialiser)
    {
        detail::check(mlpp_closest_centroid_create(centroids_initialiser ? centroids_initialiser->handle() : nullptr, &h_));
    }
private:
    std::shared_ptr<const CentroidsInitialiser> centroids_initialiser_;
};

/** Abstract clustering model (ML/Clustering.hpp:17-55). */
class Model {
public:
    virtual ~Model() {}
    virtual bool fit(Eigen::Ref<const Eigen::MatrixXd> data) = 0;
    virtual unsigned int number_clusters() const = 0;
    virtual const std::vector<unsigned int>& labels() const = 0;
    virtual const Eigen::MatrixXd& centroids() const = 0;
    virtual bool converged() const = 0;
};

/** K-means (ML/KMeans.hpp:16-111). */
class KMeans : public Model {
public:
    explicit KMeans(unsigned int number_clusters) : number_clusters_(number_clusters) { detail::check(mlpp_kmeans_create(number_clusters, &h_)); }
    ~KMeans() override { if (h_) mlpp_kmeans_destroy(h_); }
    KMeans(const KMeans&) = delete;
    KMeans& operator=(const KMeans&) = delete;

    bool fit(Eigen::Ref<const Eigen::MatrixXd> data) override
    {
        const detail::Block block(data);
        int converged = 0;
        detail::check(mlpp_kmeans_fit(h_, block.p, static_cast<uint64_t>(data.cols()), static_cast<uint32_t>(data.rows()), &converged));
        centroids_.resize(data.rows(), number_clusters_);
        detail::check(mlpp_kmeans_centroids(h_, centroids_.data()));
        labels_.resize(static_cast<std::size_t>(data.cols()));
        static_assert(sizeof(unsigned int) == sizeof(uint32_t), "labels are 32-bit");
        detail::check(mlpp_kmeans_labels(h_, reinterpret_cast<uint32_t*>(labels_.data())));
        detail::check(mlpp_kmeans_inertia(h_, &inertia_));
        converged_ = converged != 0;
        return converged_;
    }
    unsigned int number_clusters() const override { return number_clusters_; }
    const std::vector<unsigned int>& labels() const override { return labels_; }
    const Eigen::MatrixXd& centroids() const override { return centroids_; }
    void set_seed(unsigned int seed) { detail::check(mlpp_kmeans_set_seed(h_, seed)); }
    void set_absolute_tolerance(double absolute_tolerance) { detail::check(mlpp_kmeans_set_absolute_tolerance(h_, absolute_tolerance)); }
    void set_maximum_steps(unsigned int maximum_steps) { detail::check(mlpp_kmeans_set_maximum_steps(h_, maximum_steps)); }
    void set_number_initialisations(unsigned int number_initialisations) { detail::check(mlpp_kmeans_set_number_initialisations(h_, number_initialisations)); }
    void set_centroids_initialiser(std::shared_ptr<const CentroidsInitialiser> centroids_initialiser)
    {
        detail::check(mlpp_kmeans_set_centroids_initialiser(h_, centroids_initialiser ? centroids_initialiser->handle() : nullptr));
        centroids_initialiser_ = centroids_initialiser;
    }
    void set_verbose(bool verbose) { detail::check(mlpp_kmeans_set_verbose(h_, verbose ? 1 : 0)); }
    /** (label, squared distance) of the nearest centroid (ML/KMeans.cpp:153-165). */
    std::pair<unsigned int, double> assign_label(Eigen::Ref<const Eigen::VectorXd> x) const
    {
        uint32_t label = 0;
        double dist2 = 0;
        const Eigen::VectorXd contiguous(x);
        detail::check(mlpp_kmeans_assign_label(h_, contiguous.data(), static_cast<uint32_t>(contiguous.size()), &label, &dist2));
        return std::make_pair(static_cast<unsigned int>(label), dist2);
    }
    double inertia() const { return inertia_; }
    bool converged() const override { return converged_; }

private:
    mlpp_kmeans* h_ = nullptr;
    std::shared_ptr<const CentroidsInitialiser> centroids_initialiser_;
    Eigen::MatrixXd centroids_;
    std::vector<unsigned int> labels_;
    double inertia_ = 0;
    unsigned int number_clusters_;
    bool converged_ = false;
};

}  // namespace Clustering

/** Gaussian-mixture EM (ML/EM.hpp:18-198). */
class EM : public Clustering::Model {
public:
    explicit EM(unsigned int number_components) : number_components_(number_components), covariances_(number_components)
    {
        detail::check(mlpp_em_create(number_components, &h_));
    }
    ~EM() override { if (h_) mlpp_em_destroy(h_); }
    EM(const EM&) = delete;
    EM& operator=(const EM&) = delete;

    void set_seed(unsigned int seed) { detail::check(mlpp_em_set_seed(h_, seed)); }
    void set_absolute_tolerance(double absolute_tolerance) { detail::check(mlpp_em_set_absolute_tolerance(h_, absolute_tolerance)); }
    void set_relative_tolerance(double relative_tolerance) { detail::check(mlpp_em_set_relative_tolerance(h_, relative_tolerance)); }
    void set_maximum_steps(unsigned int maximum_steps) { detail::check(mlpp_em_set_maximum_steps(h_, maximum_steps)); }
    void set_means_initialiser(std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser)
    {
        detail::check(mlpp_em_set_means_initialiser(h_, means_initialiser ? means_initialiser->handle() : nullptr));
        means_initialiser_ = means_initialiser;
    }
    void set_responsibilities_initialiser(std::shared_ptr<const Clustering::ResponsibilitiesInitialiser> responsibilities_initialiser)
    {
        detail::check(mlpp_em_set_responsibilities_initialiser(h_, responsibilities_initialiser ? responsibilities_initialiser->handle() : nullptr));
        responsibilities_initialiser_ = responsibilities_initialiser;
    }
    void set_verbose(bool verbose) { detail::check(mlpp_em_set_verbose(h_, verbose ? 1 : 0)); }
    void set_maximise_first(bool maximise_first) { detail::check(mlpp_em_set_maximise_first(h_, maximise_first ? 1 : 0)); }

    bool fit(Eigen::Ref<const Eigen::MatrixXd> data) override
    {
        const detail::Block block(data);
        int converged = 0;
        detail::check(mlpp_em_fit(h_, block.p, static_cast<uint64_t>(data.cols()), static_cast<uint32_t>(data.rows()), &converged));
        const Eigen::Index d = data.rows();
        means_.resize(d, number_components_);
        detail::check(mlpp_em_means(h_, means_.data()));
        mixing_probabilities_.resize(number_components_);
        detail::check(mlpp_em_mixing_probabilities(h_, mixing_probabilities_.data()));
        for (unsigned int k = 0; k < number_components_; ++k) {
            covariances_[k].resize(d, d);
            detail::check(mlpp_em_covariance(h_, k, covariances_[k].data()));
        }
        labels_.resize(static_cast<std::size_t>(data.cols()));
        static_assert(sizeof(unsigned int) == sizeof(uint32_t), "labels are 32-bit");
        detail::check(mlpp_em_labels(h_, reinterpret_cast<uint32_t*>(labels_.data())));
        detail::check(mlpp_em_log_likelihood(h_, &log_likelihood_));
        converged_ = converged != 0;
        responsibilities_fetched_ = false;
        return converged_;
    }

    unsigned int number_components() const { return number_components_; }
    unsigned int number_clusters() const override { return number_components_; }
    const Eigen::MatrixXd& means() const { return means_; }
    const Eigen::MatrixXd& centroids() const override { return means_; }
    const std::vector<Eigen::MatrixXd>& covariances() const { return covariances_; }
    /** @throw std::invalid_argument If `k >= number_components()` (ML/EM.cpp:84-89). */
    const Eigen::MatrixXd& covariance(unsigned int k) const
    {
        if (k >= number_components_) throw std::invalid_argument("EM: Bad component index");
        return covariances_[k];
    }
    const Eigen::VectorXd& mixing_probabilities() const { return mixing_probabilities_; }
    /** N x number_components(); copied from the device on the first call after a fit. */
    const Eigen::MatrixXd& responsibilities() const
    {
        if (!responsibilities_fetched_) {
            responsibilities_.resize(static_cast<Eigen::Index>(labels_.size()), number_components_);
            detail::check(mlpp_em_responsibilities(h_, responsibilities_.data()));
            responsibilities_fetched_ = true;
        }
        return responsibilities_;
    }
    double log_likelihood() const { return log_likelihood_; }
    std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser() const { return means_initialiser_; }
    /** @throw std::invalid_argument If `x.size() != means().rows()` or `u.size() != number_components()` (ML/EM.cpp:178-183). */
    void assign_responsibilities(Eigen::Ref<const Eigen::VectorXd> x, Eigen::Ref<Eigen::VectorXd> u) const
    {
        const Eigen::VectorXd contiguous(x);
        Eigen::VectorXd out(u.size());
        detail::check(mlpp_em_assign_responsibilities(h_, contiguous.data(), static_cast<uint32_t>(contiguous.size()), out.data(),
                                                      static_cast<uint32_t>(out.size())));
        u = out;
    }
    const std::vector<unsigned int>& labels() const override { return labels_; }
    bool converged() const override { return converged_; }

private:
    mlpp_em* h_ = nullptr;
    std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser_;
    std::shared_ptr<const Clustering::ResponsibilitiesInitialiser> responsibilities_initialiser_;
    unsigned int number_components_;
    Eigen::MatrixXd means_;
    std::vector<Eigen::MatrixXd> covariances_;
    Eigen::VectorXd mixing_probabilities_;
    mutable Eigen::MatrixXd responsibilities_;
    mutable bool responsibilities_fetched_ = false;
    std::vector<unsigned int> labels_;
    double log_likelihood_ = 0;
    bool converged_ = false;
};

}  // namespace eigen_api
}  // namespace ml
