// ml::EM -- driver loop of Gaussian-mixture EM with the behaviour of the reference's ML/EM.cpp:17-188; the
// E-step / M-step bodies (ML/EM.cpp:190-263) and the label pass (:289-304) execute on the GPU through mlhip.h.
#include "ML/EM.hpp"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <typeinfo>

#include "DeviceInit.hpp"
#include "ML/Device.hpp"
#include "ML/LinearAlgebra.hpp"
#include "mlhip.h"

namespace ml {

using device::check;

namespace {
/// d x K matrix of means -> the ABI wants exactly this memory (column k = mean k).
struct DataGuard {
    mlhip_data* h = nullptr;
    ~DataGuard() { if (h) mlhip_data_free(h); }
    mlhip_data* release() { mlhip_data* t = h; h = nullptr; return t; }
};

void print_row(const double* p, Index n, Index stride)
{
    for (Index j = 0; j < n; ++j) std::cout << (j ? " " : "") << p[j * stride];
}
}  // namespace

EM::EM(const unsigned int number_components)
    : means_initialiser_(std::make_shared<Clustering::Forgy>())
    , responsibilities_initialiser_(std::make_shared<Clustering::ClosestCentroid>(means_initialiser_))
    , mixing_probabilities_(number_components)
    , covariances_(number_components)
    , inverse_covariances_(number_components)
    , sqrt_covariance_determinants_(number_components)
    , absolute_tolerance_(1e-8)
    , relative_tolerance_(1e-8)
    , log_likelihood_(0)
    , number_components_(number_components)
    , maximum_steps_(1000)
    , steps_done_(0)
    , verbose_(false)
    , maximise_first_(false)
    , converged_(false)
{
    if (!number_components) throw std::invalid_argument("EM: At least one component required");
}

EM::~EM()
{
    if (device_data_) mlhip_data_free(device_data_);
}

void EM::release_device_data()
{
    if (device_data_) {
        if (responsibilities_on_device_) {
            try { (void)responsibilities(); } catch (...) {}
        }
        mlhip_data_free(device_data_);
        device_data_ = nullptr;
    }
    responsibilities_on_device_ = false;
}

void EM::set_seed(unsigned int seed) { prng_.seed(seed); }

void EM::set_absolute_tolerance(double absolute_tolerance)
{
    if (absolute_tolerance < 0) throw std::domain_error("EM: Negative absolute tolerance");
    absolute_tolerance_ = absolute_tolerance;
}

void EM::set_relative_tolerance(double relative_tolerance)
{
    if (relative_tolerance < 0) throw std::domain_error("EM: Negative relative tolerance");
    relative_tolerance_ = relative_tolerance;
}

void EM::set_maximum_steps(unsigned int maximum_steps)
{
    if (maximum_steps < 2) throw std::invalid_argument("EM: At least two steps required for convergence test");
    maximum_steps_ = maximum_steps;
}

void EM::set_means_initialiser(std::shared_ptr<const Clustering::CentroidsInitialiser> means_initialiser)
{
    if (!means_initialiser) throw std::invalid_argument("EM: Null means initialiser");
    means_initialiser_ = means_initialiser;
}

void EM::set_responsibilities_initialiser(std::shared_ptr<const Clustering::ResponsibilitiesInitialiser> responsibilities_initialiser)
{
    if (!responsibilities_initialiser) throw std::invalid_argument("EM: Null responsibilities initialiser");
    responsibilities_initialiser_ = responsibilities_initialiser;
}

const MatrixXd& EM::covariance(unsigned int k) const
{
    if (k >= number_components_) throw std::invalid_argument("EM: Bad component index");
    return covariances_[k];
}

const MatrixXd& EM::responsibilities() const
{
    if (responsibilities_on_device_ && device_data_) {
        // Lazy device -> host copy of the N x K block (column-major, like the reference's member); the host matrix
        // (N*K doubles, 5 GB at the headline size) is only allocated now.
        responsibilities_.resize(static_cast<Index>(labels_.size()), number_components_);
        check(mlhip_em_responsibilities(device::context(), device_data_, number_components_, responsibilities_.data(),
                                        responsibilities_.rows()));
        responsibilities_on_device_ = false;
    }
    return responsibilities_;
}

MatrixXd EM::responsibilities_rows(Index first_row, Index number_rows) const
{
    const Index n = static_cast<Index>(labels_.size());
    if (first_row < 0 || number_rows < 0 || first_row > n || number_rows > n - first_row)
        throw std::invalid_argument("EM: Row range beyond the sample");
    MatrixXd out(number_rows, number_components_);
    if (responsibilities_on_device_ && device_data_) {
        check(mlhip_em_responsibilities_rows(device::context(), device_data_, number_components_, static_cast<uint64_t>(first_row),
                                             static_cast<uint64_t>(number_rows), out.data(), std::max<Index>(number_rows, 1)));
    } else {
        for (unsigned int k = 0; k < number_components_; ++k)
            std::copy_n(responsibilities_.col(k) + first_row, number_rows, out.col(k));
    }
    return out;
}

bool EM::fit(ConstMatrixRef data)
{
    converged_ = false;
    steps_done_ = 0;
    const auto number_dimensions = static_cast<unsigned int>(data.rows());
    const auto sample_size = static_cast<unsigned int>(data.cols());
    const unsigned int K = number_components_;
    if (!number_dimensions) throw std::invalid_argument("EM: At least one dimension required");

    // A previous fit's device block is no longer needed (its responsibilities are superseded).
    responsibilities_on_device_ = false;
    release_device_data();

    // The exact fit (N == K) and the "not enough data" test are about the WHOLE sample: in a row-sharded multi-rank job
    // (a hook or a communicator installed on the facade's context, which therefore exists already) the local shard may
    // be smaller than K, even empty, and every rank must still take the same branch and join the same collectives.
    int world = 1, rank = 0;
    mlhip_ctx* ctx = device::peek_context();
    if (ctx) check(mlhip_ctx_world(ctx, &world, &rank));
    if (world == 1) {
        if (sample_size < K) throw std::invalid_argument("EM: Not enough data ");
        if (sample_size > K) ctx = device::context();     // (N == K needs no device at all)
    }

    means_.resize(number_dimensions, K);
    mixing_probabilities_.fill(1. / static_cast<double>(K));
    labels_.resize(sample_size);

    if (world == 1 && sample_size == K) {
        // An exact deterministic fit: one Gaussian per sample, zero variance (ML/EM.cpp:108-118). Host only.
        responsibilities_.setZero(sample_size, sample_size);
        for (unsigned int i = 0; i < sample_size; ++i) {
            responsibilities_(i, i) = 1;
            std::copy_n(data.col(i), number_dimensions, means_.col(i));
            covariances_[i].setZero(number_dimensions, number_dimensions);
            labels_[i] = i;
        }
        log_likelihood_ = std::numeric_limits<double>::infinity();
        converged_ = true;
        return converged_;
    }

    // ---- move the sample block to HBM (stays resident for the whole fit) ---------------------------------
    DataGuard dev;
    check(mlhip_data_upload(ctx, data.data(), number_dimensions, sample_size, data.outerStride(), &dev.h));
    uint64_t n_global = 0;
    check(mlhip_data_shape(dev.h, nullptr, nullptr, &n_global));
    if (n_global < K) throw std::invalid_argument("EM: Not enough data ");       // the same on every rank
    if (n_global == K) {
        // The exact fit of a row-sharded sample: component (first_row + i) is this rank's sample i; the means are
        // put together across ranks, everything else is local (ML/EM.cpp:108-118).
        Index first_row = 0, total = 0;
        Clustering::detail::locate_rows(ctx, sample_size, first_row, total);
        responsibilities_.setZero(sample_size, K);
        means_.setZero();
        for (unsigned int i = 0; i < sample_size; ++i) {
            const auto g = static_cast<unsigned int>(first_row) + i;
            responsibilities_(i, g) = 1;
            std::copy_n(data.col(i), number_dimensions, means_.col(g));
            labels_[i] = g;
        }
        Clustering::detail::sum_across_ranks(ctx, means_);
        for (unsigned int k = 0; k < K; ++k) covariances_[k].setZero(number_dimensions, number_dimensions);
        log_likelihood_ = std::numeric_limits<double>::infinity();
        converged_ = true;
        return converged_;
    }

    const std::size_t dd = static_cast<std::size_t>(number_dimensions) * number_dimensions;
    const bool diagonal = covariance_type_ == CovarianceType::Diagonal;   // extension, see ML/EM.hpp
    std::vector<double> cov_flat(dd * K);
    std::vector<double> var_flat;                                          // diagonal mode: K x d variances
    auto unpack_covariances = [&] {
        for (unsigned int k = 0; k < K; ++k) {
            covariances_[k].resize(number_dimensions, number_dimensions);
            std::copy_n(cov_flat.data() + dd * k, dd, covariances_[k].data());
        }
    };

    if (maximise_first_) {
        // Start from responsibilities, then one M-step (ML/EM.cpp:120-125).
        const auto* closest = dynamic_cast<const Clustering::ClosestCentroid*>(responsibilities_initialiser_.get());
        const Clustering::ResponsibilitiesInitialiser& initialiser = *responsibilities_initialiser_;
        if (closest && typeid(initialiser) == typeid(Clustering::ClosestCentroid)) {
            // Library ClosestCentroid: draw the centroids on the host exactly as it would, run the nearest-centroid
            // pass (strict '<', first minimum wins: ML/Clustering.cpp:77-88) on the GPU and accumulate the one-hot
            // M-step from labels -- no N x K matrix is materialised.
            MatrixXd centroids(number_dimensions, K);
            Clustering::detail::init_centroids(*closest->centroids_initialiser(), data, prng_, K, centroids, ctx, dev.h);
            double inertia = 0;
            uint64_t changed = 0;
            check(mlhip_kmeans_assign(ctx, dev.h, K, centroids.data(), &inertia, &changed));
            std::vector<unsigned int> hard(sample_size);
            check(mlhip_kmeans_labels(ctx, dev.h, hard.data()));
            check(mlhip_em_maximisation_from_labels(ctx, dev.h, K, hard.data(), mixing_probabilities_.data(), means_.data(),
                                                    cov_flat.data()));
        } else {
            // User-defined initialiser: it fills a host N x K matrix, which is shipped to the device once.
            responsibilities_.resize(sample_size, K);
            responsibilities_initialiser_->init(data, prng_, K, responsibilities_);
            check(mlhip_em_maximisation_from(ctx, dev.h, K, responsibilities_.data(), responsibilities_.rows(),
                                             mixing_probabilities_.data(), means_.data(), cov_flat.data()));
        }
    } else {
        // Sensible guesses: initialiser's means, every covariance = sample covariance (ML/EM.cpp:127-135).
        Clustering::detail::init_centroids(*means_initialiser_, data, prng_, K, means_, ctx, dev.h);   // identical on all ranks
        std::vector<double> sample_covariance(dd);
        check(mlhip_sample_covariance(ctx, dev.h, nullptr, sample_covariance.data()));
        for (unsigned int k = 0; k < K; ++k) std::copy_n(sample_covariance.data(), dd, cov_flat.data() + dd * k);
    }

    if (diagonal) {
        // the diagonal of the starting covariances (a one-hot / responsibility-weighted M-step entry by entry, or the
        // sample variances) is the diagonal-mode start
        var_flat.resize(static_cast<std::size_t>(number_dimensions) * K);
        for (unsigned int k = 0; k < K; ++k)
            for (unsigned int j = 0; j < number_dimensions; ++j)
                var_flat[static_cast<std::size_t>(k) * number_dimensions + j] = cov_flat[dd * k + static_cast<std::size_t>(j) * number_dimensions + j];
    }

    if (!verbose_) {
        // The whole loop below in one library call (same steps, same convergence test: ML/EM.cpp:143-170), with the M-step's
        // closing arithmetic and the K covariance factorizations on the device between two tests.
        uint32_t steps = 0;
        int conv = 0;
        check(mlhip_em_iterate(ctx, dev.h, K, diagonal ? MLHIP_COVARIANCE_DIAGONAL : MLHIP_COVARIANCE_FULL,
                               mixing_probabilities_.data(), means_.data(), diagonal ? var_flat.data() : cov_flat.data(),
                               maximum_steps_, absolute_tolerance_, relative_tolerance_, &steps, &conv, &log_likelihood_, nullptr));
        steps_done_ = steps;
        if (conv) {
            check(mlhip_em_labels(ctx, dev.h, K, labels_.data()));   // EM::calculate_labels, on convergence only
            converged_ = true;
        }
    }

    double old_log_likelihood = -std::numeric_limits<double>::infinity();
    for (unsigned int step = 0; verbose_ && step < maximum_steps_; ++step) {
        // One E-step + M-step on the device; parameters are updated in place (ML/EM.cpp:145-147).
        if (diagonal)
            check(mlhip_em_step_diag(ctx, dev.h, K, mixing_probabilities_.data(), means_.data(), var_flat.data(), &log_likelihood_,
                                     mixing_probabilities_.data(), means_.data(), var_flat.data()));
        else
            check(mlhip_em_step(ctx, dev.h, K, mixing_probabilities_.data(), means_.data(), cov_flat.data(), &log_likelihood_,
                                mixing_probabilities_.data(), means_.data(), cov_flat.data()));
        ++steps_done_;

        if (verbose_) {
            std::cout << "Step " << step << "\n";
            std::cout << "Log-likelihood == " << log_likelihood_ << "\n";
            std::cout << "Mixing probabilities == ";
            print_row(mixing_probabilities_.data(), K, 1);
            std::cout << "\n";
            for (unsigned int k = 0; k < K; ++k) {
                std::cout << "Mean[" << k << "] == ";
                print_row(means_.col(k), number_dimensions, 1);
                std::cout << "\n";
            }
            std::cout << "Responsibilities (first 10 rows):\n";
            // (only the rows that are printed cross the bus: ML/EM.cpp:155-156 prints topRows(10))
            MatrixXd r(std::min(sample_size, 10u), K);
            check(mlhip_em_responsibilities_rows(ctx, dev.h, K, 0, static_cast<uint64_t>(r.rows()), r.data(), r.rows()));
            for (unsigned int i = 0; i < std::min(sample_size, 10u); ++i) {
                print_row(r.data() + i, K, r.rows());
                std::cout << "\n";
            }
            std::cout << std::endl;
        }

        if (step > 0) {
            const double ll_change = std::abs(log_likelihood_ - old_log_likelihood);
            if (ll_change < absolute_tolerance_ + relative_tolerance_ * std::max(std::abs(old_log_likelihood), std::abs(log_likelihood_))) {
                check(mlhip_em_labels(ctx, dev.h, K, labels_.data()));   // EM::calculate_labels, on convergence only
                converged_ = true;
                break;
            }
        }
        old_log_likelihood = log_likelihood_;
    }

    if (diagonal) {
        std::fill(cov_flat.begin(), cov_flat.end(), 0.0);
        for (unsigned int k = 0; k < K; ++k)
            for (unsigned int j = 0; j < number_dimensions; ++j)
                cov_flat[dd * k + static_cast<std::size_t>(j) * number_dimensions + j] = var_flat[static_cast<std::size_t>(k) * number_dimensions + j];
    }
    unpack_covariances();
    process_covariances(number_dimensions);
    // The responsibilities of the last E-step stay in HBM; responsibilities() allocates and fetches them on demand.
    responsibilities_.resize(0, 0);
    responsibilities_on_device_ = true;
    device_data_ = dev.release();
    return converged_;
}

void EM::assign_responsibilities(ConstVectorRef x, VectorRef u) const
{
    if (x.size() != means().rows()) throw std::invalid_argument("Wrong x size");
    if (u.size() != static_cast<Index>(number_components())) throw std::invalid_argument("Wrong u size");
    if (inverse_covariances_.empty() || inverse_covariances_[0].rows() != x.size())
        throw std::invalid_argument("EM: model has no fitted covariance decompositions");
    std::vector<double> diff(static_cast<std::size_t>(x.size()));
    double total = 0;
    for (unsigned int k = 0; k < number_components_; ++k) {
        const double* mean = means_.col(k);
        for (Index j = 0; j < x.size(); ++j) diff[static_cast<std::size_t>(j)] = x[j] - mean[j];
        const double q = LinearAlgebra::xAx_symmetric(inverse_covariances_[k], diff);
        u[k] = std::exp(-0.5 * q) * mixing_probabilities_[k] / sqrt_covariance_determinants_[k];
        total += u[k];
    }
    for (unsigned int k = 0; k < number_components_; ++k) u[k] /= total;
}

void EM::process_covariances(const Index number_dimensions)
{
    // Inverse and sqrt(det) of every covariance for the host point query (ML/EM.cpp:274-287).
    for (unsigned int k = 0; k < number_components_; ++k) {
        inverse_covariances_[k].resize(number_dimensions, number_dimensions);
        check(mlhip_process_covariance(static_cast<uint32_t>(number_dimensions), covariances_[k].data(),
                                       inverse_covariances_[k].data(), &sqrt_covariance_determinants_[k]));
    }
}

}  // namespace ml
