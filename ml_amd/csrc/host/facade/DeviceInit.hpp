// Internal to the facade: initialisers whose O(N K d) distance passes run on the GPU-resident copy of the data
// (SURVEY.md section 8 f1). Results are bit-identical to the host implementations in Clustering.cpp: the per-sample
// squared distance is the same fma chain on both sides, minima are exact, and the random draws happen on the host with the
// same libstdc++ calls.
#pragma once
#include <random>

#include "ML/Clustering.hpp"

struct mlhip_ctx;
struct mlhip_data;

namespace ml {
namespace Clustering {
namespace detail {

/// Runs `initialiser` for `data`; when it is exactly the library's KPP and the context is single-rank, the
/// nearest-chosen-centroid distances (ML/Clustering.cpp:44-51) are computed on the device incrementally
/// (min with the distance to the newest centroid: one N*d pass per centroid instead of N*n*d).
void init_centroids(const CentroidsInitialiser& initialiser, ConstMatrixRef data, std::default_random_engine& prng,
                    unsigned int number_components, MatrixRef centroids, mlhip_ctx* ctx, mlhip_data* device_data);

/// Row-sharded job: global index of this rank's first row and the global row count (rank-ordered shards; one short
/// all-reduce). Single rank: 0 and n_local.
void locate_rows(mlhip_ctx* ctx, Index n_local, Index& first_row, Index& n_global);

/// Sums a d x K block across ranks in place (each rank contributes its own columns, zeros elsewhere).
void sum_across_ranks(mlhip_ctx* ctx, MatrixRef m);

}  // namespace detail
}  // namespace Clustering
}  // namespace ml
