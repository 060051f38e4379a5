// ml::Clustering::KMeans -- driver of Lloyd's algorithm with the behaviour of the reference's ML/KMeans.cpp:10-151;
// assignment_step / update_step (ML/KMeans.cpp:153-192) execute on the GPU through mlhip.h.
#include "ML/KMeans.hpp"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <stdexcept>

#include "DeviceInit.hpp"
#include "ML/Device.hpp"
#include "mlhip.h"

namespace ml {
namespace Clustering {

using device::check;

namespace {
struct DataGuard {
    mlhip_data* h = nullptr;
    ~DataGuard() { if (h) mlhip_data_free(h); }
};
/// Above this many samples the final inertia is the device's tree sum instead of the host's sequential sum.
constexpr std::size_t kSequentialInertiaLimit = std::size_t(1) << 24;
}  // namespace

KMeans::KMeans(unsigned int number_clusters)
    : work_vector_(number_clusters)
    , centroids_initialiser_(std::make_shared<Clustering::Forgy>())
    , absolute_tolerance_(1e-8)
    , inertia_(0)
    , maximum_steps_(1000)
    , num_inits_(1)
    , num_clusters_(number_clusters)
    , steps_done_(0)
    , verbose_(false)
    , converged_(false)
{
    if (!number_clusters) throw std::invalid_argument("KMeans: number of clusters cannot be zero");
}

KMeans::~KMeans() {}

void KMeans::set_seed(unsigned int seed) { prng_.seed(seed); }

void KMeans::set_absolute_tolerance(double absolute_tolerance)
{
    if (absolute_tolerance < 0) throw std::domain_error("KMeans: Negative absolute tolerance");
    absolute_tolerance_ = absolute_tolerance;
}

void KMeans::set_maximum_steps(unsigned int maximum_steps)
{
    if (maximum_steps < 2) throw std::invalid_argument("KMeans: At least two steps required for convergence test");
    maximum_steps_ = maximum_steps;
}

void KMeans::set_number_initialisations(unsigned int number_initialisations)
{
    if (number_initialisations < 1) throw std::invalid_argument("KMeans: At least 1 initialisation required");
    num_inits_ = number_initialisations;
}

void KMeans::set_centroids_initialiser(std::shared_ptr<const Clustering::CentroidsInitialiser> centroids_initialiser)
{
    if (!centroids_initialiser) throw std::invalid_argument("KMeans: Null centroids initialiser");
    centroids_initialiser_ = centroids_initialiser;
}

std::pair<unsigned int, double> KMeans::assign_label(ConstVectorRef x) const
{
    // Same IEEE operations, in the same order, as the device kernel (ml_amd/csrc/device/kmeans.hip): the distance
    // returned here is bit-identical to the one the fit used for this point.
    const Index d = centroids_.rows();
    double nearest = std::numeric_limits<double>::infinity();
    unsigned int label = 0;
    for (unsigned int k = 0; k < num_clusters_; ++k) {
        const double* c = centroids_.col(k);
        double dist = 0;
        for (Index j = 0; j < d; ++j) {
            const double t = x[j] - c[j];
            dist = std::fma(t, t, dist);
        }
        if (dist < nearest) { nearest = dist; label = k; }
    }
    return std::make_pair(label, nearest);
}

void KMeans::fetch_assignment(mlhip_data* device_data, std::size_t sample_size)
{
    mlhip_ctx* ctx = device::context();
    check(mlhip_kmeans_labels(ctx, device_data, labels_.data()));
    sequential_inertia(device_data, sample_size);
}

void KMeans::sequential_inertia(mlhip_data* device_data, std::size_t sample_size)
{
    mlhip_ctx* ctx = device::context();
    int world = 1;
    check(mlhip_ctx_world(ctx, &world, nullptr));
    if (world == 1 && sample_size <= kSequentialInertiaLimit) {
        // inertia_ accumulated sample by sample like ML/KMeans.cpp:172-177, from the device's per-sample distances.
        std::vector<double> dist(sample_size);
        check(mlhip_kmeans_distances(ctx, device_data, dist.data()));
        double total = 0;
        for (double v : dist) total += v;
        inertia_ = total;
    }
}

bool KMeans::fit(ConstMatrixRef data)
{
    const auto number_dimensions = static_cast<unsigned int>(data.rows());
    const auto sample_size = static_cast<unsigned int>(data.cols());
    if (!number_dimensions) throw std::invalid_argument("KMeans: At least one dimension required");

    // "Not enough data" and the exact fit are about the WHOLE sample: in a row-sharded multi-rank job (hook or
    // communicator installed on the facade's context, which therefore exists already) a shard may hold fewer than K
    // rows, or none, and every rank must still take the same branch and join the same collectives.
    DataGuard dev;
    int world = 1;
    mlhip_ctx* ctx = device::peek_context();
    if (ctx) check(mlhip_ctx_world(ctx, &world, nullptr));
    uint64_t n_global = sample_size;
    if (world == 1) {
        if (sample_size < num_clusters_) throw std::invalid_argument("KMeans: Not enough data ");
        if (sample_size > num_clusters_)      // (N == K needs no device at all)
            check(mlhip_data_upload(device::context(), data.data(), number_dimensions, sample_size, data.outerStride(), &dev.h));
    } else {
        check(mlhip_data_upload(ctx, data.data(), number_dimensions, sample_size, data.outerStride(), &dev.h));
        check(mlhip_data_shape(dev.h, nullptr, nullptr, &n_global));
        if (n_global < num_clusters_) throw std::invalid_argument("KMeans: Not enough data ");   // the same on every rank
    }
    const bool exact_fit = n_global == num_clusters_;

    if (num_inits_ == 1 || exact_fit) {
        const bool ok = fit_once(data, dev.h, exact_fit);
        if (dev.h && !exact_fit) fetch_assignment(dev.h, sample_size);
        return ok;
    }
    converged_ = false;
    double min_inertia = std::numeric_limits<double>::infinity();
    MatrixXd best_centroids;
    for (unsigned int i = 0; i < num_inits_; ++i) {
        if (fit_once(data, dev.h, false)) {
            if (inertia_ < min_inertia) {
                min_inertia = inertia_;
                best_centroids = centroids_;
            }
            converged_ = true;
        }
    }
    if (converged_) {
        centroids_ = best_centroids;
        if (dev.h) {
            uint64_t changed = 0;
            check(mlhip_kmeans_assign(device::context(), dev.h, num_clusters_, centroids_.data(), &inertia_, &changed));
        }
    }
    if (dev.h) fetch_assignment(dev.h, sample_size);
    return converged_;
}

bool KMeans::fit_once(ConstMatrixRef data, mlhip_data* device_data, const bool exact_fit)
{
    converged_ = false;
    steps_done_ = 0;
    const auto number_dimensions = static_cast<unsigned int>(data.rows());
    const auto sample_size = static_cast<unsigned int>(data.cols());
    const unsigned int K = num_clusters_;

    centroids_.resize(number_dimensions, K);
    old_centroids_.resize(number_dimensions, K);
    labels_.resize(sample_size);

    if (exact_fit) {
        // Exact fit: every sample is its own cluster (ML/KMeans.cpp:67-75). Host only; in a row-sharded job cluster
        // (first_row + i) is this rank's sample i and the centroid block is put together across ranks.
        Index first_row = 0, total = sample_size;
        mlhip_ctx* shared = device::peek_context();
        int ranks = 1;
        if (shared) check(mlhip_ctx_world(shared, &ranks, nullptr));
        if (ranks > 1) detail::locate_rows(shared, sample_size, first_row, total);
        centroids_.setZero();
        for (unsigned int i = 0; i < sample_size; ++i) {
            const auto g = static_cast<unsigned int>(first_row) + i;
            std::copy_n(data.col(i), number_dimensions, centroids_.col(g));
            labels_[i] = g;
        }
        if (ranks > 1) detail::sum_across_ranks(shared, centroids_);
        inertia_ = 0;
        converged_ = true;
        return converged_;
    }

    mlhip_ctx* ctx = device::context();
    int world = 1, rank = 0;
    check(mlhip_ctx_world(ctx, &world, &rank));
    detail::init_centroids(*centroids_initialiser_, data, prng_, K, centroids_, ctx, device_data);   // identical on all ranks

    if (!verbose_) {
        // The whole step loop in one call: between two stopping tests the centroid table stays on the device (same
        // arithmetic and the same decisions as the loop below, which prints the centroids of every step).
        uint32_t steps = 0;
        int converged = 0;
        check(mlhip_kmeans_iterate(ctx, device_data, K, centroids_.data(), old_centroids_.data(), maximum_steps_, absolute_tolerance_,
                                   &steps, &converged, &inertia_, work_vector_.data()));
        steps_done_ = steps;
        converged_ = converged != 0;
        sequential_inertia(device_data, sample_size);
        return converged_;
    }

    MatrixXd updated(number_dimensions, K);
    for (unsigned int step = 0; step < maximum_steps_; ++step) {
        // Assignment + the per-cluster sums of the update in one pass over the resident data.
        uint64_t changed = 0;
        check(mlhip_kmeans_step(ctx, device_data, K, centroids_.data(), &inertia_, &changed, work_vector_.data(), updated.data()));
        ++steps_done_;

        if (step > 0 && changed == 0) {   // same labels twice (ML/KMeans.cpp:84-89): centroids stay as they are
            converged_ = true;
            break;
        }

        old_centroids_.swap(centroids_);  // update_step (ML/KMeans.cpp:180-192): empty clusters sit at the origin
        centroids_ = updated;

        if (verbose_) {
            std::cout << "Step " << step << "\n";
            for (unsigned int k = 0; k < K; ++k) {
                std::cout << "Centroid[" << k << "] ==";
                for (unsigned int j = 0; j < number_dimensions; ++j) std::cout << " " << centroids_(j, k);
                std::cout << "\n";
            }
            std::cout << std::endl;
        }

        if (step > 0) {
            double centroid_shift = 0;
            for (Index t = 0; t < centroids_.size(); ++t) {
                const double delta = centroids_.data()[t] - old_centroids_.data()[t];
                centroid_shift += delta * delta;
            }
            if (centroid_shift < absolute_tolerance_) {
                check(mlhip_kmeans_assign(ctx, device_data, K, centroids_.data(), &inertia_, &changed));
                converged_ = true;
                break;
            }
        }
    }
    // The multi-initialisation loop compares inertias of different runs with '<' (ML/KMeans.cpp:35): give it the
    // reference's sequentially accumulated value, not the device's tree sum.
    sequential_inertia(device_data, sample_size);
    return converged_;
}

}  // namespace Clustering
}  // namespace ml
