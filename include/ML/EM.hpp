#pragma once
#ifdef MLHIP_ML_EIGEN_API_HPP
#error "ML/EM.hpp and ML/EigenApi.hpp share their class names: include one family per translation unit"
#endif
#define MLHIP_ML_EM_HPP
/* ml::EM -- Gaussian-mixture Expectation-Maximisation with the public interface of the reference's
 * ML/EM.hpp:18-198 (same method names, defaults, exceptions and result semantics), executed on an MI355X:
 * the E-step / M-step loops of ML/EM.cpp:190-263 run as HIP kernels through the C ABI in mlhip.h; this class keeps
 * the driver loop (ML/EM.cpp:91-174), the convergence test and the host-side point query. */
#include <memory>
#include <random>
#include <vector>

#include "Clustering.hpp"
#include "Dense.hpp"
#include "dll.hpp"

struct mlhip_data;

namespace ml {

class EM : public Clustering::Model {
public:
    /** @throw std::invalid_argument If `number_components == 0`. */
    DLL_DECLSPEC EM(unsigned int number_components);
    DLL_DECLSPEC ~EM() override;
    EM(const EM&) = delete;
    EM& operator=(const EM&) = delete;

    DLL_DECLSPEC void set_seed(unsigned int seed);
    /** @throw std::domain_error If `absolute_tolerance < 0`. */
    DLL_DECLSPEC void set_absolute_tolerance(double absolute_tolerance);
    /** @throw std::domain_error If `relative_tolerance < 0`. */
    DLL_DECLSPEC void set_relative_tolerance(double relative_tolerance);
    /** @throw std::invalid_argument If `maximum_steps < 2`. */
    DLL_DECLSPEC void set_maximum_steps(unsigned int maximum_steps);
    /** @throw std::invalid_argument If `means_initialiser` is null. */
    DLL_DECLSPEC void set_means_initialiser(std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser);
    /** @throw std::invalid_argument If `responsibilities_initialiser` is null. */
    DLL_DECLSPEC void set_responsibilities_initialiser(std::shared_ptr<const Clustering::ResponsibilitiesInitialiser> responsibilities_initialiser);
    void set_verbose(bool verbose) { verbose_ = verbose; }
    void set_maximise_first(bool maximise_first) { maximise_first_ = maximise_first; }
    /** EXTENSION -- not part of the reference API, whose EM is full-covariance only (reference ML/EM.hpp:175). With
    `Diagonal` every covariance is restricted to its diagonal (the E-/M-step loops of ML/EM.cpp:190-263 on the diagonal
    entries only; covariances() returns diagonal matrices). Default `Full` == the reference's behaviour.
    One fused kernel serves number_dimensions <= 32 and number_components <= 64; other shapes run the full-covariance kernels on
    diagonal matrices (slower, never refused). */
    enum class CovarianceType { Full, Diagonal };
    void set_covariance_type(CovarianceType covariance_type) { covariance_type_ = covariance_type; }
    CovarianceType covariance_type() const { return covariance_type_; }

    /** @brief Fits the model. @param[in] data Column-major, a data point in every column (this rank's row shard when an
    all-reduce hook is installed on the device context). @return `true` if fitting converged.
    @throw std::invalid_argument If `data` has no rows or fewer columns than components.
    @throw std::runtime_error On device failures (no GPU, HIP or collective errors). */
    DLL_DECLSPEC bool fit(ConstMatrixRef data) override;

    unsigned int number_components() const { return number_components_; }
    unsigned int number_clusters() const override { return number_components(); }
    /** `number_dimensions` x number_components(). */
    const MatrixXd& means() const { return means_; }
    const MatrixXd& centroids() const override { return means(); }
    const std::vector<MatrixXd>& covariances() const { return covariances_; }
    /** @throw std::invalid_argument If `k >= number_components()`. */
    DLL_DECLSPEC const MatrixXd& covariance(unsigned int k) const;
    const VectorXd& mixing_probabilities() const { return mixing_probabilities_; }
    /** `sample_size` x number_components(): the responsibilities of the last E-step. They stay on the device after
    fit() and are copied to the host on the first call (N x K doubles), unlike the reference which always holds them. */
    DLL_DECLSPEC const MatrixXd& responsibilities() const;
    /** Extension: rows [first_row, first_row + number_rows) of responsibilities() as a number_rows x number_components() matrix,
    WITHOUT materialising the whole block on the host when it still lives on the device (what a slice of a 5 GB block costs).
    @throw std::invalid_argument If the range exceeds the sample. */
    DLL_DECLSPEC MatrixXd responsibilities_rows(Index first_row, Index number_rows) const;
    double log_likelihood() const { return log_likelihood_; }
    std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser() const { return means_initialiser_; }
    /** Posterior component probabilities of one point under the fitted parameters (host-side).
    @throw std::invalid_argument If `x.size() != means().rows()` or `u.size() != number_components()`. */
    DLL_DECLSPEC void assign_responsibilities(ConstVectorRef x, VectorRef u) const;
    const std::vector<unsigned int>& labels() const override { return labels_; }
    bool converged() const override { return converged_; }
    /** Extension: number of E-M iterations the last fit() ran. */
    unsigned int steps_done() const { return steps_done_; }
    /** Extension: frees the HBM copy of the data kept for responsibilities(). */
    DLL_DECLSPEC void release_device_data();

private:
    std::default_random_engine prng_;
    std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser_;
    std::shared_ptr<const Clustering::ResponsibilitiesInitialiser> responsibilities_initialiser_;
    VectorXd mixing_probabilities_;
    MatrixXd means_;
    mutable MatrixXd responsibilities_;
    mutable bool responsibilities_on_device_ = false;
    std::vector<MatrixXd> covariances_;
    std::vector<MatrixXd> inverse_covariances_;
    VectorXd sqrt_covariance_determinants_;
    std::vector<unsigned int> labels_;
    double absolute_tolerance_;
    double relative_tolerance_;
    double log_likelihood_;
    unsigned int number_components_;
    unsigned int maximum_steps_;
    unsigned int steps_done_;
    bool verbose_;
    bool maximise_first_;
    bool converged_;
    CovarianceType covariance_type_ = CovarianceType::Full;
    mlhip_data* device_data_ = nullptr;

    void process_covariances(Index number_dimensions);
};

}  // namespace ml
