/* mlpp_c.h -- flat C handles over the C++ facade (the headers under include/ML), so that the Python surface
 * ml_amd.cppyml.clustering (the mirror of the reference's pybind11 module, cppyml/clustering.cpp:75-185) is plain
 * ctypes and needs neither pybind11 nor Eigen. Every function returns 0 or an MLHIP_E_* code (see mlhip.h);
 * mlhip_last_error() holds the message. std::invalid_argument / std::domain_error map to MLHIP_E_INVALID_ARGUMENT /
 * MLHIP_E_DOMAIN -- the Python side raises ValueError for both, as pybind11 does for the reference. */
#ifndef MLPP_C_H
#define MLPP_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mlpp_centroids_initialiser mlpp_centroids_initialiser;           /* shared_ptr<const CentroidsInitialiser> */
typedef struct mlpp_responsibilities_initialiser mlpp_responsibilities_initialiser;
typedef struct mlpp_em mlpp_em;
typedef struct mlpp_kmeans mlpp_kmeans;

/* ---- initialisers (cppyml/clustering.cpp:79-101) ---- */
int mlpp_forgy_create(mlpp_centroids_initialiser** out);
int mlpp_random_partition_create(mlpp_centroids_initialiser** out);
int mlpp_kpp_create(mlpp_centroids_initialiser** out);
/* Extension: fixed centroids, K x d row-major (row k = centroid k). */
int mlpp_fixed_centroids_create(const double* centroids, uint32_t K, uint32_t d, mlpp_centroids_initialiser** out);
int mlpp_centroids_initialiser_destroy(mlpp_centroids_initialiser* h);
/* Runs the initialiser on N x d row-major data with a std::default_random_engine seeded with `seed` (seed_set != 0)
 * or default-constructed; out: K x d row-major. For tests of the host-side initialisers. */
int mlpp_centroids_initialiser_run(const mlpp_centroids_initialiser* h, const double* data, uint64_t n, uint32_t d,
                                   uint32_t K, int seed_set, uint32_t seed, double* centroids_out);
/* The same through the path `fit` takes: the data is uploaded to the facade's GPU and the initialiser's O(N) passes (K-means++
 * distances, RandomPartition's running means: SURVEY 8 f1) run there; the result is bit-identical to the host run above. */
int mlpp_centroids_initialiser_run_on_device(const mlpp_centroids_initialiser* h, const double* data, uint64_t n, uint32_t d,
                                             uint32_t K, int seed_set, uint32_t seed, double* centroids_out);
int mlpp_closest_centroid_create(const mlpp_centroids_initialiser* centroids_initialiser, mlpp_responsibilities_initialiser** out);
int mlpp_responsibilities_initialiser_destroy(mlpp_responsibilities_initialiser* h);
/* out: N x K column-major. */
int mlpp_responsibilities_initialiser_run(const mlpp_responsibilities_initialiser* h, const double* data, uint64_t n,
                                          uint32_t d, uint32_t K, int seed_set, uint32_t seed, double* resp_out);

/* ---- EM (cppyml/clustering.cpp:103-146; data = N x d row-major == d x N column-major) ---- */
int mlpp_em_create(uint32_t number_components, mlpp_em** out);
int mlpp_em_destroy(mlpp_em* h);
int mlpp_em_set_seed(mlpp_em* h, uint32_t seed);
int mlpp_em_set_absolute_tolerance(mlpp_em* h, double v);
int mlpp_em_set_relative_tolerance(mlpp_em* h, double v);
int mlpp_em_set_maximum_steps(mlpp_em* h, uint32_t v);
int mlpp_em_set_means_initialiser(mlpp_em* h, const mlpp_centroids_initialiser* init);
int mlpp_em_set_responsibilities_initialiser(mlpp_em* h, const mlpp_responsibilities_initialiser* init);
int mlpp_em_set_verbose(mlpp_em* h, int v);
int mlpp_em_set_maximise_first(mlpp_em* h, int v);
/* Extension (ml::EM::set_covariance_type): 0 = full covariances (the reference), 1 = diagonal. */
int mlpp_em_set_covariance_type(mlpp_em* h, int diagonal);
int mlpp_em_fit(mlpp_em* h, const double* data, uint64_t n, uint32_t d, int* converged);
int mlpp_em_number_components(const mlpp_em* h, uint32_t* out);
int mlpp_em_dims(const mlpp_em* h, uint32_t* d, uint64_t* n);
int mlpp_em_means(const mlpp_em* h, double* out /* d x K column-major, like EM::means() */);
int mlpp_em_covariance(const mlpp_em* h, uint32_t k, double* out /* d x d */);
int mlpp_em_mixing_probabilities(const mlpp_em* h, double* out);
int mlpp_em_responsibilities(const mlpp_em* h, double* out /* N x K column-major */);
/* Extension: rows [first_row, first_row + n_rows) only (n_rows x K column-major) -- no N x K host copy while the block is on the device. */
int mlpp_em_responsibilities_rows(const mlpp_em* h, uint64_t first_row, uint64_t n_rows, double* out);
int mlpp_em_log_likelihood(const mlpp_em* h, double* out);
int mlpp_em_labels(const mlpp_em* h, uint32_t* out);
int mlpp_em_converged(const mlpp_em* h, int* out);
int mlpp_em_steps_done(const mlpp_em* h, uint32_t* out);
int mlpp_em_assign_responsibilities(const mlpp_em* h, const double* x, uint32_t xlen, double* u, uint32_t ulen);

/* ---- KMeans (cppyml/clustering.cpp:148-183) ---- */
int mlpp_kmeans_create(uint32_t number_clusters, mlpp_kmeans** out);
int mlpp_kmeans_destroy(mlpp_kmeans* h);
int mlpp_kmeans_set_seed(mlpp_kmeans* h, uint32_t seed);
int mlpp_kmeans_set_absolute_tolerance(mlpp_kmeans* h, double v);
int mlpp_kmeans_set_maximum_steps(mlpp_kmeans* h, uint32_t v);
int mlpp_kmeans_set_number_initialisations(mlpp_kmeans* h, uint32_t v);
int mlpp_kmeans_set_centroids_initialiser(mlpp_kmeans* h, const mlpp_centroids_initialiser* init);
int mlpp_kmeans_set_verbose(mlpp_kmeans* h, int v);
int mlpp_kmeans_fit(mlpp_kmeans* h, const double* data, uint64_t n, uint32_t d, int* converged);
int mlpp_kmeans_number_clusters(const mlpp_kmeans* h, uint32_t* out);
int mlpp_kmeans_dims(const mlpp_kmeans* h, uint32_t* d, uint64_t* n);
int mlpp_kmeans_centroids(const mlpp_kmeans* h, double* out /* K x d row-major == d x K column-major */);
int mlpp_kmeans_labels(const mlpp_kmeans* h, uint32_t* out);
int mlpp_kmeans_inertia(const mlpp_kmeans* h, double* out);
int mlpp_kmeans_converged(const mlpp_kmeans* h, int* out);
int mlpp_kmeans_steps_done(const mlpp_kmeans* h, uint32_t* out);
int mlpp_kmeans_assign_label(const mlpp_kmeans* h, const double* x, uint32_t xlen, uint32_t* label, double* dist2);

/* ---- LinearAlgebra helpers (ML/LinearAlgebra.hpp:18-31), column-major ---- */
int mlpp_xAx_symmetric(const double* A, uint32_t rows, uint32_t cols, const double* x, uint32_t xlen, double* out);
int mlpp_xxT(const double* x, uint32_t n, double* dest /* n x n */);
int mlpp_add_a_xxT(const double* x, uint32_t n, double* dest, uint32_t drows, uint32_t dcols, double a);

/* ---- LinearRegression::calculate_XXt_beta (ML/LinearRegression.hpp:412): X is N x q row-major == q x N column-major ---- */
int mlpp_calculate_XXt_beta(const double* X, uint64_t n, uint32_t q, const double* y, uint64_t ylen, const double* lambda,
                            uint32_t lambda_len, double* XXt /* q x q */, double* beta /* q */);

/* ---- device context of the facade (include/ML/Device.hpp): the process-wide mlhip_ctx the model classes run on. A
 * row-sharded job installs its all-reduce hook on it (mlhip_ctx_set_allreduce) before calling fit() on every rank. ---- */
int mlpp_device_context(struct mlhip_ctx** out);          /* ml::device::context(): created on first use, not owned by the caller */
int mlpp_device_set_context(struct mlhip_ctx* ctx);       /* ml::device::set_context(); NULL restores the default */

#ifdef __cplusplus
}
#endif
#endif /* MLPP_C_H */
