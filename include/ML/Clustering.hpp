#pragma once
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
/** @brief Methods and classes for clustering algorithms. */
namespace Clustering {

/** @brief Abstract clustering model. */
class Model {
public:
    DLL_DECLSPEC virtual ~Model();
    /** @brief Fits the model. @param[in] data Column-major matrix with a data point in every column.
    @return `true` if fitting converged. @throw std::invalid_argument If `data` has no rows or too few columns. */
    virtual bool fit(ConstMatrixRef data) = 0;
    virtual unsigned int number_clusters() const = 0;
    virtual const std::vector<unsigned int>& labels() const = 0;
    virtual const MatrixXd& centroids() const = 0;
    virtual bool converged() const = 0;
};

/** @brief Chooses initial locations of centroids. */
class CentroidsInitialiser {
public:
    DLL_DECLSPEC virtual ~CentroidsInitialiser();
    /** @param[out] centroids `data.rows()` x `number_components`. */
    DLL_DECLSPEC virtual void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const = 0;
};

/** @brief Chooses initial component responsibilities. */
class ResponsibilitiesInitialiser {
public:
    DLL_DECLSPEC virtual ~ResponsibilitiesInitialiser();
    /** @param[out] responsibilities `data.cols()` x `number_components`. */
    DLL_DECLSPEC virtual void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef responsibilities) const = 0;
};

/** @brief Chooses random points as new centroids. */
class Forgy : public CentroidsInitialiser {
public:
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const override;
};

/** @brief Assigns points to clusters randomly and then returns cluster means. */
class RandomPartition : public CentroidsInitialiser {
public:
    DLL_DECLSPEC void init(ConstMatrixRef data, std::default_random_engine& prng, unsigned int number_components, MatrixRef centroids) const override;
};

/** @brief K-means++ seeding. */
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

/** @brief Initialises centroids and then assigns the responsibility for each point to its closest centroid. */
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
