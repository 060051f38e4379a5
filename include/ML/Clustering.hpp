#pragma once
#ifdef MLHIP_ML_EIGEN_API_HPP
#error "ML/Clustering.hpp and ML/EigenApi.hpp share their class names: include one family per translation unit"
#endif
#define MLHIP_ML_CLUSTERING_HPP
/* Interfaces and initialisers of the clustering models: same class names, virtual signatures and semantics as
 * the reference's ML/Clustering.hpp:17-127 (Eigen argument types replaced by the views of Dense.hpp).
 * Initialisers run on the host on the caller's data, exactly like the reference (same libstdc++ <random> calls,
 * so the same seed gives the same draw); they are user-subclassable extension points. */
#include <memory>
#include <random>
#include <vector>

#include "Dense.hpp"
#include "dll.hpp"

namespace ml {
namespace Clustering {

/** What every clustering estimator of this library offers once fitted (pure interface). */
class Model {
public:
    DLL_DECLSPEC virtual ~Model();
    /** Estimates the model from `data` (d x N, one sample per column); the block is only borrowed for the call.
    Returns whether the iteration converged; throws std::invalid_argument for d == 0 or fewer samples than clusters. */
    virtual bool fit(ConstMatrixRef data) = 0;
    virtual unsigned int number_clusters() const = 0;
    virtual const std::vector<unsigned int>& labels() const = 0;
    virtual const MatrixXd& centroids() const = 0;
    virtual bool converged() const = 0;
};

/** Strategy object: produces the K starting centroids of a fit. Subclass it to plug in your own. */
class CentroidsInitialiser {
public:
    DLL_DECLSPEC virtual ~CentroidsInitialiser();
    /** Fills `centroids` (d x K, one centroid per column) from the d x N sample block, drawing from `prng` if it needs randomness. */
    DLL_DECLSPEC virtual void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const = 0;
};

/** Strategy object: produces the N x K starting responsibilities of an EM fit that begins with an M-step. */
class ResponsibilitiesInitialiser {
public:
    DLL_DECLSPEC virtual ~ResponsibilitiesInitialiser();
    /** Fills `responsibilities` (N x K, rows sum to one). */
    DLL_DECLSPEC virtual void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef responsibilities) const = 0;
};

#ifdef MLHIP_HAVE_EIGEN
/** Base class for user initialisers written against the REFERENCE signature (reference ML/Clustering.hpp:71): derive from
this instead of CentroidsInitialiser and keep the Eigen-typed `init` as it is. (The library itself is built without Eigen,
so its own virtual takes the views of Dense.hpp; this adapter, emitted entirely in the user's translation unit, forwards.) */
class EigenCentroidsInitialiser : public CentroidsInitialiser {
public:
    virtual void init(Eigen::Ref<const Eigen::MatrixXd> data, std::default_random_engine& prng, unsigned int number_components,
                      Eigen::Ref<Eigen::MatrixXd> centroids) const = 0;
    void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const final
    {
        using Stride = Eigen::OuterStride<>;
        const Eigen::Map<const Eigen::MatrixXd, 0, Stride> in(data.data(), data.rows(), data.cols(), Stride(data.outerStride()));
        const Eigen::Map<Eigen::MatrixXd, 0, Stride> out(centroids.data(), centroids.rows(), centroids.cols(), Stride(centroids.outerStride()));
        // exact argument types: a Map would convert to the views of the other overload just as well
        const Eigen::Ref<const Eigen::MatrixXd> in_ref(in);
        const Eigen::Ref<Eigen::MatrixXd> out_ref(out);
        init(in_ref, prng, number_components, out_ref);
    }
};

/** The same for responsibilities initialisers (reference ML/Clustering.hpp:88). */
class EigenResponsibilitiesInitialiser : public ResponsibilitiesInitialiser {
public:
    virtual void init(Eigen::Ref<const Eigen::MatrixXd> data, std::default_random_engine& prng, unsigned int number_components,
                      Eigen::Ref<Eigen::MatrixXd> responsibilities) const = 0;
    void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef responsibilities) const final
    {
        using Stride = Eigen::OuterStride<>;
        const Eigen::Map<const Eigen::MatrixXd, 0, Stride> in(data.data(), data.rows(), data.cols(), Stride(data.outerStride()));
        const Eigen::Map<Eigen::MatrixXd, 0, Stride> out(responsibilities.data(), responsibilities.rows(), responsibilities.cols(),
                                                         Stride(responsibilities.outerStride()));
        const Eigen::Ref<const Eigen::MatrixXd> in_ref(in);
        const Eigen::Ref<Eigen::MatrixXd> out_ref(out);
        init(in_ref, prng, number_components, out_ref);
    }
};
#endif

/** K distinct samples, picked uniformly (selection sampling: ascending sample indices). */
class Forgy : public CentroidsInitialiser {
public:
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const override;
};

/** Every sample gets a uniformly random cluster; the centroids are the (running) means of those random groups. */
class RandomPartition : public CentroidsInitialiser {
public:
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const override;
};

/** K-means++: each new centroid is a sample drawn with probability proportional to its squared distance from the
nearest centroid chosen so far (the first one uniformly). */
class KPP : public CentroidsInitialiser {
public:
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const override;
};

/** @brief Extension (not in the reference): returns the centroids it was constructed with. */
class FixedCentroids : public CentroidsInitialiser {
public:
    /** @param centroids `number_dimensions` x `number_components`. */
    DLL_DECLSPEC explicit FixedCentroids(const MatrixXd& centroids);
    /** @throw std::invalid_argument If the stored centroids do not match `data.rows()` x `number_components`. */
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const override;
private:
    MatrixXd centroids_;
};

/** One-hot responsibilities: asks a CentroidsInitialiser for centroids, then gives each sample to its nearest one
(first minimum wins). */
class ClosestCentroid : public ResponsibilitiesInitialiser {
public:
    /** @throw std::invalid_argument If `centroids_initialiser` is null. */
    DLL_DECLSPEC ClosestCentroid(std::shared_ptr<const CentroidsInitialiser> centroids_initialiser);
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef responsibilities) const override;
    /** The centroids initialiser (lets ml::EM run the nearest-centroid pass on the GPU). */
    std::shared_ptr<const CentroidsInitialiser> centroids_initialiser() const { return centroids_initialiser_; }
private:
    std::shared_ptr<const CentroidsInitialiser> centroids_initialiser_;
};

}  // namespace Clustering
}  // namespace ml
