#include "DeviceInit.hpp"

#include <algorithm>
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
    // K-means++ (ML/Clustering.cpp:39-59) with the distance passes on the device.
    const Index n = data.cols(), d = data.rows();
    std::vector<double> weights(static_cast<std::size_t>(n), 1.0), latest;
    for (unsigned int chosen = 0; chosen < number_components; ++chosen) {
        if (chosen > 0) {
            // squared distance of every sample to the centroid chosen last; weights = running minimum
            std::vector<double>& target = chosen == 1 ? weights : latest;
            target.resize(static_cast<std::size_t>(n));
            device::check(mlhip_min_squared_distances(ctx, device_data, 1, centroids.col(chosen - 1), target.data()));
            if (chosen > 1)
                for (std::size_t i = 0; i < weights.size(); ++i) weights[i] = std::min(weights[i], latest[i]);
        }
        std::discrete_distribution<Index> draw(weights.begin(), weights.end());
        std::copy_n(data.col(draw(prng)), d, centroids.col(chosen));
    }
}

}  // namespace detail
}  // namespace Clustering
}  // namespace ml
