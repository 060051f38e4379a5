#pragma once
#ifdef MLHIP_ML_EIGEN_API_HPP
#error "ML/KMeans.hpp and ML/EigenApi.hpp share their class names: include one family per translation unit"
#endif
#define MLHIP_ML_KMEANS_HPP
/* ml::Clustering::KMeans -- Lloyd K-means with the public interface of the reference's ML/KMeans.hpp:18-125,
 * executed on an MI355X: assignment_step/update_step (ML/KMeans.cpp:153-192) run as one HIP kernel per step through
 * the C ABI in mlhip.h; this class keeps fit/fit_once (ML/KMeans.cpp:25-114) and the host-side point query. */
#include <memory>
#include <random>
#include <utility>
#include <vector>

#include "Clustering.hpp"
#include "Dense.hpp"
#include "dll.hpp"

struct mlhip_data;

namespace ml {
namespace Clustering {

class KMeans : public Model {
public:
    /** @throw std::invalid_argument If `number_clusters == 0`. */
    DLL_DECLSPEC KMeans(unsigned int number_clusters);
    DLL_DECLSPEC ~KMeans() override;
    KMeans(const KMeans&) = delete;
    KMeans& operator=(const KMeans&) = delete;

    DLL_DECLSPEC bool fit(ConstMatrixRef data) override;
    unsigned int number_clusters() const override { return num_clusters_; }
    const std::vector<unsigned int>& labels() const override { return labels_; }
    const MatrixXd& centroids() const override { return centroids_; }
    DLL_DECLSPEC void set_seed(unsigned int seed);
    /** @throw std::domain_error If `absolute_tolerance < 0`. */
    DLL_DECLSPEC void set_absolute_tolerance(double absolute_tolerance);
    /** @throw std::invalid_argument If `maximum_steps < 2`. */
    DLL_DECLSPEC void set_maximum_steps(unsigned int maximum_steps);
    /** @throw std::invalid_argument If `number_initialisations < 1`. */
    DLL_DECLSPEC void set_number_initialisations(unsigned int number_initialisations);
    /** @throw std::invalid_argument If `centroids_initialiser` is null. */
    DLL_DECLSPEC void set_centroids_initialiser(std::shared_ptr<const CentroidsInitialiser> centroids_initialiser);
    void set_verbose(bool verbose) { verbose_ = verbose; }
    /** Label and squared Euclidean distance of the nearest centroid (host-side point query). */
    DLL_DECLSPEC std::pair<unsigned int, double> assign_label(ConstVectorRef x) const;
    double inertia() const { return inertia_; }
    bool converged() const override { return converged_; }
    /** Extension: number of Lloyd steps the last fit_once ran. */
    unsigned int steps_done() const { return steps_done_; }

private:
    std::vector<unsigned int> labels_;
    MatrixXd centroids_;
    MatrixXd old_centroids_;
    VectorXd work_vector_;
    std::default_random_engine prng_;
    std::shared_ptr<const CentroidsInitialiser> centroids_initialiser_;
    double absolute_tolerance_;
    double inertia_;
    unsigned int maximum_steps_;
    unsigned int num_inits_;
    unsigned int num_clusters_;
    unsigned int steps_done_;
    bool verbose_;
    bool converged_;

    /// exact_fit: the WHOLE sample has exactly K rows (one cluster per sample, ML/KMeans.cpp:67-75).
    bool fit_once(ConstMatrixRef data, mlhip_data* device_data, bool exact_fit);
    void fetch_assignment(mlhip_data* device_data, std::size_t sample_size);
    void sequential_inertia(mlhip_data* device_data, std::size_t sample_size);
};

}  // namespace Clustering
}  // namespace ml
