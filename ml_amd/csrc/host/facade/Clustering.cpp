// Initialisers with the behaviour of the reference's ML/Clustering.cpp:16-89: same libstdc++ <random> calls in the
// same order on the same value types, so a given seed produces the same draw as the reference built with libstdc++.
#include "ML/Clustering.hpp"

#include <algorithm>
#include <cmath>
#include <iterator>
#include <limits>
#include <numeric>
#include <stdexcept>

namespace ml {
namespace Clustering {

namespace {
inline double squared_distance(const double* x, const double* c, Index d)
{
    double s = 0;
    for (Index j = 0; j < d; ++j) {
        const double t = x[j] - c[j];
        s = std::fma(t, t, s);   // the same fma chain as the device kernel (ml_amd/csrc/device/kmeans.hip)
    }
    return s;
}
}  // namespace

Model::~Model() {}
CentroidsInitialiser::~CentroidsInitialiser() {}
ResponsibilitiesInitialiser::~ResponsibilitiesInitialiser() {}

void Forgy::init(ConstMatrixRef data, std::default_random_engine& prng, const unsigned int number_components, MatrixRef centroids) const
{
    // K distinct sample indices by selection sampling (ascending order), ML/Clustering.cpp:18-21.
    std::vector<Index> candidates(static_cast<std::size_t>(data.cols()));
    std::iota(candidates.begin(), candidates.end(), 0);
    std::vector<Index> chosen;
    std::sample(candidates.begin(), candidates.end(), std::back_inserter(chosen), number_components, prng);
    for (unsigned int k = 0; k < number_components; ++k)
        std::copy_n(data.col(chosen[k]), data.rows(), centroids.col(k));
}

void RandomPartition::init(ConstMatrixRef data, std::default_random_engine& prng, const unsigned int number_components, MatrixRef centroids) const
{
    centroids.setZero();
    std::vector<unsigned int> sizes(number_components, 0);
    std::uniform_int_distribution<unsigned int> pick(0, number_components - 1);
    const Index d = data.rows();
    for (Index i = 0; i < data.cols(); ++i) {
        const unsigned int k = pick(prng);
        const double count = static_cast<double>(++sizes[k]);
        double* c = centroids.col(k);
        const double* x = data.col(i);
        for (Index j = 0; j < d; ++j) c[j] += (x[j] - c[j]) / count;   // running mean, ML/Clustering.cpp:34
    }
}

void KPP::init(ConstMatrixRef data, std::default_random_engine& prng, const unsigned int number_components, MatrixRef centroids) const
{
    const Index n = data.cols(), d = data.rows();
    std::vector<double> weights(static_cast<std::size_t>(n));
    for (unsigned int chosen = 0; chosen < number_components; ++chosen) {
        if (chosen == 0) {
            std::fill(weights.begin(), weights.end(), 1);
        } else {
            for (Index i = 0; i < n; ++i) {
                double nearest = std::numeric_limits<double>::infinity();
                for (unsigned int k = 0; k < chosen; ++k)
                    nearest = std::min(nearest, squared_distance(data.col(i), centroids.col(k), d));
                weights[static_cast<std::size_t>(i)] = nearest;
            }
        }
        std::discrete_distribution<Index> draw(weights.begin(), weights.end());
        std::copy_n(data.col(draw(prng)), d, centroids.col(chosen));
    }
}

FixedCentroids::FixedCentroids(const MatrixXd& centroids) : centroids_(centroids) {}

void FixedCentroids::init(ConstMatrixRef data, std::default_random_engine&, const unsigned int number_components, MatrixRef centroids) const
{
    if (centroids_.rows() != data.rows() || centroids_.cols() != static_cast<Index>(number_components))
        throw std::invalid_argument("FixedCentroids: stored centroids do not match the requested shape");
    for (Index k = 0; k < centroids_.cols(); ++k) std::copy_n(centroids_.col(k), centroids_.rows(), centroids.col(k));
}

ClosestCentroid::ClosestCentroid(std::shared_ptr<const CentroidsInitialiser> centroids_initialiser)
    : centroids_initialiser_(centroids_initialiser)
{
    if (!centroids_initialiser) throw std::invalid_argument("Null centroids initialiser");
}

void ClosestCentroid::init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef responsibilities) const
{
    MatrixXd centroids(data.rows(), number_components);
    centroids_initialiser_->init(data, prng, number_components, centroids);
    responsibilities.setZero();
    const Index d = data.rows();
    for (Index i = 0; i < data.cols(); ++i) {
        double nearest = squared_distance(data.col(i), centroids.col(0), d);
        unsigned int arg = 0;
        for (unsigned int k = 1; k < number_components; ++k) {
            const double dist = squared_distance(data.col(i), centroids.col(k), d);
            if (dist < nearest) { nearest = dist; arg = k; }   // strict '<': first minimum wins
        }
        responsibilities(i, arg) = 1;
    }
}

}  // namespace Clustering
}  // namespace ml
