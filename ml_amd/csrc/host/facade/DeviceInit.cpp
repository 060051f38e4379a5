#include "DeviceInit.hpp"

#include <algorithm>
#include <limits>
#include <random>
#include <typeinfo>
#include <vector>

#include "ML/Device.hpp"
#include "mlhip.h"

namespace ml {
namespace Clustering {
namespace detail {

void init_centroids(const CentroidsInitialiser& initialiser, ConstMatrixRef data, std::default_random_engine& prng,
                    const unsigned int number_components, MatrixRef centroids, mlhip_ctx* ctx, mlhip_data* device_data)
{
    int world = 1;
    if (ctx) device::check(mlhip_ctx_world(ctx, &world, nullptr));
    if (!ctx || !device_data || world != 1 || typeid(initialiser) != typeid(KPP)) {
        initialiser.init(data, prng, number_components, centroids);
        return;
    }
    // K-means++ (ML/Clustering.cpp:39-59) with the distance passes on the device. The draw follows what
    // std::discrete_distribution (libstdc++ bits/random.tcc, _M_initialize + operator()) computes for these weights -- the
    // sequential sum, p_i = w_i / sum, sequential cumulative sums, first index whose cumulative probability is >= the
    // canonical uniform draw, last one forced to 1 -- without materialising the two N-long probability vectors: the
    // cumulative scan stops at the drawn index.
    const Index n = data.cols(), d = data.rows();
    const std::size_t count = static_cast<std::size_t>(n);
    std::vector<double> weights(count, 1.0), latest;
    for (unsigned int chosen = 0; chosen < number_components; ++chosen) {
        double sum = 0.0;
        if (chosen == 0) {
            sum = static_cast<double>(count);     // == the sequential sum of `count` ones (exact below 2^53)
        } else {
            // squared distance of every sample to the centroid chosen last; weights = running minimum
            std::vector<double>& target = chosen == 1 ? weights : latest;
            target.resize(count);
            device::check(mlhip_min_squared_distances(ctx, device_data, 1, centroids.col(chosen - 1), target.data()));
            if (chosen == 1) {
                for (std::size_t i = 0; i < count; ++i) sum += weights[i];
            } else {
                for (std::size_t i = 0; i < count; ++i) {
                    const double w = std::min(weights[i], latest[i]);
                    weights[i] = w;
                    sum += w;
                }
            }
        }
        std::size_t pick = 0;
        if (count >= 2) {
            const double p = std::generate_canonical<double, std::numeric_limits<double>::digits>(prng);
            double cumulative = 0.0;
            pick = count - 1;
            for (std::size_t i = 0; i + 1 < count; ++i) {
                cumulative += weights[i] / sum;
                if (cumulative >= p) { pick = i; break; }
            }
        }
        std::copy_n(data.col(static_cast<Index>(pick)), d, centroids.col(chosen));
    }
}

}  // namespace detail
}  // namespace Clustering
}  // namespace ml
