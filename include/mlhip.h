/* mlhip.h -- C ABI of the MI355X (gfx950) Gaussian-mixture EM / K-means hot path.
 *
 * Drop-in boundary. The reference (romanwerpachowski/ML, "ML++") has no FFI of its own for this
 * path: the hot loops are private C++ member functions. Each entry point below is the device-side
 * replacement of one of them and cites the reference function it stands in for (paths relative to
 * the reference checkout). The C++ facade in include/ML/ (ml::EM, ml::Clustering::KMeans -- same
 * names and semantics as ML/EM.hpp, ML/KMeans.hpp) and the Python surface ml_amd.cppyml.clustering
 * (same names as cppyml/clustering.cpp) are built on these calls only.
 *
 * Conventions
 *  - plain C types only; every function returns 0 on success, <0 on error (MLHIP_E_*);
 *    mlhip_last_error() gives the thread-local message of the last failure;
 *  - matrices are column-major, one sample per column (d x N), exactly the memory of the reference's
 *    `Eigen::Ref<const Eigen::MatrixXd> data` (ML/Clustering.hpp:28-33) == a C-contiguous N x d numpy
 *    array (cppyml/clustering.cpp:27-30). `ld` = distance in doubles between consecutive samples;
 *  - all pointers are HOST pointers owned by the caller unless a name ends in `_dev`;
 *  - one context drives one GPU (one process per GPU) -- or, created as a device GROUP (mlhip_ctx_create_group), all the GPUs of
 *    the node from one process; a context is not thread-safe;
 *  - there is no CPU fallback: without a usable HIP device every compute entry point fails with
 *    MLHIP_E_NO_DEVICE.
 */
#ifndef MLHIP_H
#define MLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLHIP_OK 0
#define MLHIP_E_INVALID_ARGUMENT (-1) /* maps to std::invalid_argument in the facade */
#define MLHIP_E_DOMAIN (-2)           /* maps to std::domain_error */
#define MLHIP_E_RUNTIME (-3)          /* HIP / collective failure: std::runtime_error */
#define MLHIP_E_NO_DEVICE (-4)        /* no usable GPU: std::runtime_error */
#define MLHIP_E_UNSUPPORTED (-5)      /* shape outside what the kernels are built for */

typedef struct mlhip_ctx mlhip_ctx;   /* one GPU + stream + scratch */
typedef struct mlhip_data mlhip_data; /* a d x N sample block resident in HBM (row shard of this rank) */

const char* mlhip_last_error(void);
const char* mlhip_version(void);

/* ---- context ------------------------------------------------------------------------------------ */
int mlhip_device_count(int* count);
/* device_id < 0: take MLHIP_DEVICE, else LOCAL_RANK, else 0. */
int mlhip_ctx_create(int device_id, mlhip_ctx** out);
/* DEVICE GROUP: one context over n_shards shards, shard s on GPU device_ids[s] (NULL: s mod the number of visible GPUs; an id may
 * repeat -- several shards on one GPU is how a one-GPU box runs the 8-GPU configurations at full size). This is the single-process
 * multi-GPU form of the reference's API -- `bool EM::fit(Eigen::Ref<const MatrixXd>)` (ML/EM.cpp:91) and KMeans::fit (ML/KMeans.cpp:25)
 * take ONE d x N block in ONE process: mlhip_data_upload on a group row-shards the caller's block over the shards (contiguous, balanced),
 * and EVERY entry point below accepts the group's context with the caller's WHOLE arrays (all N rows of labels / responsibilities /
 * targets / weights; results are those of a single context, up to the summation order of the statistics). Inside, one host thread per
 * shard drives that shard's GPU with the ordinary per-context code, and the shards' statistics meet once per iteration:
 *   - shards on distinct GPUs: RCCL (ncclCommInitAll, ncclAllReduce on each shard's stream);
 *   - otherwise, or with MLHIP_GROUP_REDUCE=direct: an in-process all-reduce -- every shard sums all shards' buffers in shard order with
 *     the same kernel (peer access over xGMI between GPUs), so the shards hold bit-identical sums and runs are reproducible.
 * To its caller a group is ONE rank (mlhip_ctx_world: 1, 0); it takes no all-reduce hook or communicator of its own. */
int mlhip_ctx_create_group(int n_shards, const int* device_ids, mlhip_ctx** out);
/* What the C++ facade / Python surface use (ml::device::context()): a group when the environment asks for one -- MLHIP_DEVICES=0,1,2,3
 * (one shard per entry) or MLHIP_NUM_GPUS=n (ignored under a one-process-per-GPU launcher, i.e. when LOCAL_RANK is set) -- else
 * mlhip_ctx_create(-1). */
int mlhip_ctx_create_default(mlhip_ctx** out);
int mlhip_ctx_destroy(mlhip_ctx* ctx);
/* Shards of a context (1 for an ordinary one), the GPU of a shard, and how the statistics are summed: "none", "hook-host",
 * "hook-device", "rccl" (mlhip_ctx_init_rccl), "group-rccl", "group-direct" (a string constant). */
int mlhip_ctx_shards(const mlhip_ctx* ctx, int* n_shards);
int mlhip_ctx_shard_device(const mlhip_ctx* ctx, int shard, int* device_id);
int mlhip_ctx_reduce_kind(const mlhip_ctx* ctx, const char** kind);
int mlhip_ctx_synchronize(mlhip_ctx* ctx);
int mlhip_ctx_device(const mlhip_ctx* ctx, int* device_id);
/* The HIP stream (hipStream_t) every kernel of this context is launched on. */
int mlhip_ctx_stream(const mlhip_ctx* ctx, void** stream);

/* Row-sharded multi-GPU: the N samples are split across `world_size` ranks; the only
 * exchange is one sum all-reduce per iteration of the sufficient statistics. The library calls `fn` on the
 * buffer to be summed in place across ranks; with on_device != 0 `buf` is a device pointer valid on the
 * context's stream (hand it to RCCL / torch.distributed "nccl"), else a host pointer (gloo, MPI).
 * The hook must return only when `buf` holds the global sum and is safe to read from the context's stream.
 * fn == NULL restores single-rank behaviour. */
typedef int (*mlhip_allreduce_fn)(void* user, double* buf, size_t count, int on_device, void* stream);
int mlhip_ctx_set_allreduce(mlhip_ctx* ctx, mlhip_allreduce_fn fn, void* user, int on_device,
                            int world_size, int rank);

/* Native RCCL (no Python, no hook to write): the library opens its own communicator on the context's GPU and sums the
 * statistics with ncclAllReduce(ncclDouble, ncclSum) on the context's stream -- what a multi-GPU ml::EM::fit /
 * ml::Clustering::KMeans::fit (ML/EM.hpp:82, ML/KMeans.hpp:36) uses from C++. One process per GPU:
 *   rank 0:   mlhip_rccl_unique_id(id);  ...hand the 128 bytes to the other ranks (file, pipe, MPI_Bcast, a store)...
 *   all:      mlhip_ctx_init_rccl(ctx, id, world_size, rank);        (collective: returns when all ranks have joined)
 * mlhip_ctx_init_rccl_file does the hand-over through a file that all ranks can see (rank 0 writes it atomically,
 * the others wait for it up to MLHIP_RCCL_TIMEOUT_S seconds, default 120); use a fresh path per job.
 * librccl.so.1 is loaded on first use (override: MLHIP_RCCL_LIBRARY). RCCL refuses two ranks on the same GPU.
 * Rank 0 removes the rendezvous file once the communicator exists; files older than MLHIP_RCCL_STALE_S seconds (default 600:
 * the left-over of a job that died before that) are ignored by the waiting ranks, and so is a file whose job nonce differs (a hash
 * of MLHIP_RCCL_NONCE, TORCHELASTIC_RUN_ID, MASTER_ADDR, MASTER_PORT as the ranks see them -- the same within a launch).
 * mlhip_rccl_available() says whether librccl can be loaded in THIS process without touching a GPU: ranks of a job can agree on
 * it (over whatever channel they have) BEFORE anyone enters the collective mlhip_ctx_init_rccl*, where a rank that cannot load
 * the library would leave the others waiting. */
#define MLHIP_RCCL_UNIQUE_ID_BYTES 128
int mlhip_rccl_available(void);   /* 1 / 0; never fails */
int mlhip_rccl_unique_id(void* unique_id /* MLHIP_RCCL_UNIQUE_ID_BYTES bytes */);
int mlhip_ctx_init_rccl(mlhip_ctx* ctx, const void* unique_id, int world_size, int rank);
int mlhip_ctx_init_rccl_file(mlhip_ctx* ctx, const char* path, int world_size, int rank);
/* Number of ranks of the context's own RCCL communicator as RCCL reports it (ncclCommCount); 0 without one. */
int mlhip_ctx_rccl_ranks(const mlhip_ctx* ctx, int* nranks);
int mlhip_ctx_finalize_rccl(mlhip_ctx* ctx);

/* In-place sum of `count` host doubles across ranks through the installed hook (no-op without one). The facade
 * uses it to agree on initial parameters (rank 0 contributes them, the others contribute zeros). */
int mlhip_ctx_allreduce(mlhip_ctx* ctx, double* buf, size_t count);
int mlhip_ctx_world(const mlhip_ctx* ctx, int* world_size, int* rank);

/* ---- resident data ------------------------------------------------------------------------------ */
/* Copies this rank's d x n block to HBM (stored dimension-major for coalesced per-sample access) and
 * computes the statistics shift (global column mean; all-reduced when a hook is set). The host block is
 * only read during the call (the reference borrows `data` for the duration of fit, ML/EM.cpp:91).
 * 1 <= d <= 4096 (MLHIP_E_UNSUPPORTED above; 128 < d <= 1024: matrix-core kernels of device/big_dim.hip; above: plain, untuned kernels, device/generic_dim.hip), n < 2^32 - 256 per rank. */
int mlhip_data_upload(mlhip_ctx* ctx, const double* x, uint32_t d, uint64_t n, int64_t ld, mlhip_data** out);
/* Same, from a sample-major block already in device memory (e.g. a torch tensor's data_ptr()). */
int mlhip_data_upload_dev(mlhip_ctx* ctx, const double* x_dev, uint32_t d, uint64_t n, int64_t ld, mlhip_data** out);
int mlhip_data_free(mlhip_data* data);
int mlhip_data_shape(const mlhip_data* data, uint32_t* d, uint64_t* n_local, uint64_t* n_global);
/* Rows [first_row, first_row + n_rows) of the uploaded block that shard `shard` holds (an ordinary context: shard 0, all rows). */
int mlhip_data_shard_rows(const mlhip_data* data, int shard, uint64_t* first_row, uint64_t* n_rows);
/* Global column means used as the numerical shift of the second-moment accumulation (d doubles). */
int mlhip_data_shift(const mlhip_data* data, double* shift);

/* ---- Gaussian-mixture EM -------------------------------------------------------------------------- */
/* One EM iteration == EM::expectation_step + EM::maximisation_step (ML/EM.cpp:190-263, incl.
 * process_covariances :274-287) on the resident shard, statistics all-reduced across ranks.
 *   in : mixing[K], means[d*K] (column k = mean k), covariances[K*d*d] (symmetric, column-major each)
 *   out: *log_likelihood  = mean_i log sum_k pi_k N(x_i|mu_k,Sigma_k)  under the INPUT parameters (:211)
 *        mixing_out/means_out/covariances_out = the M-step result (:229-257, ridge 1e-15 included)
 * Output arrays may alias the input arrays. The E-step results stay available on the device for
 * mlhip_em_responsibilities / mlhip_em_labels (for small shapes, where the iteration runs as one fused kernel, the
 * N x K block is rebuilt from the same parameter records when one of them is called). */
int mlhip_em_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K,
                  const double* mixing, const double* means, const double* covariances,
                  double* log_likelihood, double* mixing_out, double* means_out, double* covariances_out);

/* EXTENSION (no counterpart in the reference, whose ml::EM is full-covariance only, ML/EM.hpp:175; BASELINE.json configs[1]):
 * one EM iteration with DIAGONAL covariances -- the loops of mlhip_em_step restricted to the diagonal, in one kernel
 * (X read once, no N x K block in HBM). variances / variances_out: K*d doubles, variances[k*d + j] = sigma_kj^2 (ridge 1e-15
 * included on output, ML/EM.cpp:252). One fused kernel for d <= 32, K <= 64; other shapes run the full-covariance kernels on diagonal matrices. mlhip_em_responsibilities /
 * mlhip_em_labels afterwards work as after mlhip_em_step (the block is rebuilt from the same parameters on demand). */
int mlhip_em_step_diag(mlhip_ctx* ctx, mlhip_data* data, uint32_t K,
                       const double* mixing, const double* means, const double* variances,
                       double* log_likelihood, double* mixing_out, double* means_out, double* variances_out);

/* The iteration loop of EM::fit (ML/EM.cpp:143-170) in ONE call: up to max_steps trips of E-step + M-step, each followed by the
 * reference's convergence test |ll - ll_old| < absolute_tolerance + relative_tolerance * max(|ll_old|, |ll|) from the second
 * trip on (:161-168). Parameters are updated IN PLACE (covariances: K*d*d doubles, or K*d variances with
 * MLHIP_COVARIANCE_DIAGONAL); *log_likelihood is that of the last E-step (i.e. under the parameters before the last M-step, like
 * the reference). Between two tests everything stays on the device -- statistics, all-reduce, the M-step's closing arithmetic
 * and the K Cholesky / inverse factorizations of EM::process_covariances (:274-287), the next E-step's records -- and the
 * host reads back 1 + 2K doubles per iteration (d <= 1024: one wave per component with the matrices in LDS up to d = 64, panelled
 * factorizations in global memory above; beyond d = 1024, or with MLHIP_DEVICE_CLOSE=0, the loop runs through mlhip_em_step). Same results as calling mlhip_em_step in a loop, to the last bits of log().
 * log_likelihood_history (max_steps doubles) may be NULL. Tolerances 0 run exactly max_steps iterations. */
#define MLHIP_COVARIANCE_FULL 0
#define MLHIP_COVARIANCE_DIAGONAL 1
int mlhip_em_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, int covariance_type,
                     double* mixing, double* means, double* covariances,
                     uint32_t max_steps, double absolute_tolerance, double relative_tolerance,
                     uint32_t* steps_done, int* converged, double* log_likelihood, double* log_likelihood_history);

/* E-step only (ML/EM.cpp:190-219): leaves log-responsibilities on the device, returns the log-likelihood. */
int mlhip_em_expectation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K,
                         const double* mixing, const double* means, const double* covariances,
                         double* log_likelihood);
/* M-step only (ML/EM.cpp:221-263) from the responsibilities left by the last E-step. */
int mlhip_em_maximisation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K,
                          double* mixing_out, double* means_out, double* covariances_out);
/* M-step from caller-given responsibilities (n_local x K column-major, ldr >= n_local): the
 * `maximise_first` start (ML/EM.cpp:120-125). */
int mlhip_em_maximisation_from(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* resp, int64_t ldr,
                               double* mixing_out, double* means_out, double* covariances_out);
/* M-step from hard labels (one-hot responsibilities, what ClosestCentroid::init produces,
 * ML/Clustering.cpp:72-89) without materialising N x K on the host. */
int mlhip_em_maximisation_from_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* labels,
                                      double* mixing_out, double* means_out, double* covariances_out);

/* Normalised responsibilities of the last E-step, this rank's n_local x K block, column-major
 * (EM::responsibilities(), ML/EM.hpp:132-135; rows /= row sum, ML/EM.cpp:214-218). */
int mlhip_em_responsibilities(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* resp, int64_t ldr);
/* Rows [first_row, first_row + n_rows) of that block only (n_rows x K, column-major, ldr >= n_rows): what a verbose fit prints
 * (`responsibilities_.topRows(10)`, ML/EM.cpp:155-156) and what a Python slice of the lazy property needs -- 10 rows cost 10 rows,
 * not the N x K block. */
int mlhip_em_responsibilities_rows(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint64_t first_row, uint64_t n_rows, double* resp,
                                   int64_t ldr);
/* argmax_k of those responsibilities, first maximum wins (EM::calculate_labels, ML/EM.cpp:289-304). */
int mlhip_em_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint32_t* labels);

/* (X - mean)(X - mean)^T / (N - 1) over ALL ranks' samples (EM::calculate_sample_covariance,
 * ML/EM.cpp:265-272). mean may be NULL. */
int mlhip_sample_covariance(mlhip_ctx* ctx, mlhip_data* data, double* mean, double* covariance);

/* X X^T (d x d, column-major) and X y (d) over ALL ranks' samples in one pass over the resident block: the dense
 * contraction of LinearRegression::calculate_XXt_beta (ML/LinearRegression.cpp:213-215), the "next" row f4 of SURVEY.md
 * section 8. `y` holds this rank's n_local target values. Runs on the statistics kernel (two weight rows: 1 and y). */
int mlhip_xxt_xy(mlhip_ctx* ctx, mlhip_data* data, const double* y, double* xxt, double* xy);

/* Host helpers, no GPU needed: the layout of the all-reduced sufficient statistics and the M-step closing
 * arithmetic applied to them (ML/EM.cpp:242, 250-257). Per component, F = (d+1)(d+2)/2 doubles: the packed lower
 * triangle (row-major: entry (a,b), a >= b, at a(a+1)/2 + b) of  sum_i r_ik xt_i xt_i^T  with
 * xt_i = [x_i - shift ; 1]; so S0 = (d,d), S1'_b = (d,b), M2'_ab = (a,b). `statistics` holds K such records. */
int mlhip_em_statistics_count(uint32_t d, uint32_t* count_per_component);
int mlhip_em_finalize_statistics(uint32_t d, uint32_t K, const double* statistics, const double* shift,
                                 double n_global, double* mixing_out, double* means_out, double* covariances_out);

/* Host helper, no GPU needed: covariance -> what EM::process_covariances (ML/EM.cpp:274-287) derives:
 * inverse (d*d), sqrt(det). Used by the facade for point queries (EM::assign_responsibilities). */
int mlhip_process_covariance(uint32_t d, const double* covariance, double* inverse, double* sqrt_det);

/* ---- K-means -------------------------------------------------------------------------------------- */
/* One Lloyd step == KMeans::assignment_step + the sums KMeans::update_step needs (ML/KMeans.cpp:167-192),
 * all-reduced across ranks.
 *   in : centroids[d*K]
 *   out: *inertia = sum_i min_k |x_i - c_k|^2 ; *n_changed = #labels differing from the previous call on
 *        this data (first call: n_global); counts[K]; centroids_out[d*K] = per-cluster means, an EMPTY cluster's
 *        centroid is the origin (:184). Labels stay on the device for mlhip_kmeans_labels. */
int mlhip_kmeans_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids,
                      double* inertia, uint64_t* n_changed, double* counts, double* centroids_out);
/* The step loop of KMeans::fit_once (ML/KMeans.cpp:80-110) in ONE call: up to max_steps trips of mlhip_kmeans_step with the
 * reference's two stopping rules -- the same labels as in the previous trip (:84-89; the centroids then stay as they are), or
 * from the second trip on a squared centroid shift |C - C_old|_F^2 below absolute_tolerance, followed by one more assignment
 * (:103-108). Between two tests everything stays on the device: the update's sums are all-reduced there, divided by the counts
 * (empty cluster -> origin, :184) and become the next trip's centroid table without a host round trip; the host reads back
 * 2 + K (d + 1) doubles per trip for the two tests. Same results, bit for bit, as the loop over mlhip_kmeans_step.
 *   in/out: centroids[d*K] (start -> final); out: old_centroids[d*K] (the table before the last update; may be NULL),
 *   counts[K] of the last update (may be NULL), *inertia of the last assignment, *steps_done, *converged. */
int mlhip_kmeans_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* centroids, double* old_centroids,
                         uint32_t max_steps, double absolute_tolerance, uint32_t* steps_done, int* converged,
                         double* inertia, double* counts);
/* Assignment only (KMeans::assignment_step, :167-178): labels + inertia, no update. */
int mlhip_kmeans_assign(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids,
                        double* inertia, uint64_t* n_changed);
int mlhip_kmeans_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t* labels);
/* Per-sample squared distance to the assigned centroid from the last assignment (n_local doubles): the very values
 * KMeans::assign_label returns (:153-165), so a caller can re-create the reference's sequential inertia sum (:176). */
int mlhip_kmeans_distances(mlhip_ctx* ctx, mlhip_data* data, double* dist2);
/* One draw of KPP::init (ML/Clustering.cpp:44-58) on the resident block(s): the weights become min(weights, |x_i - centroid|^2)
 * (first != 0: the distances themselves), and the row that std::discrete_distribution returns for the canonical uniform draw u is
 * located from tree-summed cumulative weights with a rigorous error bound. Row-sharded jobs: every rank calls it with the same
 * centroid and u and the global index of its first row (first_row; 0 on a single rank); the ranks exchange their weight sums and
 * candidates through the context's all-reduce. *certain != 0: *index is that row of the WHOLE sample, bit for bit what the
 * sequential sums of the reference give (the same on every rank). *certain == 0 (two rows closer than the bound, or a zero /
 * non-finite weight sum): this rank's weights have been copied to weights_out (n_local doubles; may be NULL) for the caller's
 * sequential evaluation. At least two rows in the whole sample. */
int mlhip_kpp_draw(mlhip_ctx* ctx, mlhip_data* data, const double* centroid, int first, double u, uint64_t first_row, uint64_t* index,
                   int* certain, double* weights_out);
/* The running-minimum weights the draws of mlhip_kpp_draw have left on the device (n_local doubles): fetched by the caller only when
 * a draw was not certain -- a certified draw never moves them (ML/Clustering.cpp:44-51 keeps them in a host vector). */
int mlhip_kpp_weights(mlhip_ctx* ctx, mlhip_data* data, double* weights_out);
/* min_k |x_i - c_k|^2 per sample of this rank's shard (the weights of KPP::init, ML/Clustering.cpp:44-51). */
int mlhip_min_squared_distances(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* dist2);

/* The running means of RandomPartition::init (ML/Clustering.cpp:27-37) over this rank's rows, bit-identical to the host loop
 * `c_k += (x_i - c_k) / ++n_k`: the caller makes the per-row cluster draws (the reference's std::uniform_int_distribution calls)
 * and passes their stable partition -- order[offsets[k] .. offsets[k+1]) = the local row indices drawn for cluster k, ascending;
 * offsets[0] = 0, offsets[K] = n_local. means (K x d, centroid k = d contiguous doubles: the reference's column-major d x K) and
 * sizes (K counts) are CONTINUED: zeros for a fresh start, the previous rank's result in a row-sharded job (ranks in row order). */
int mlhip_random_partition_means(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* order, const uint32_t* offsets,
                                 double* means, double* sizes);

/* ---- timing (for bench.py / profiling) -------------------------------------------------------------- */
/* Average device time in ms of the named kernel family over its launches since the last reset, measured with
 * HIP events on the context's stream. name: "em_estep", "em_mstats", "kmeans_assign", ... Returns count in *launches.
 * While enabled, every launch is bracketed by a pair of events that is only recorded; the times are read (one stream
 * synchronisation) when mlhip_timing_get / _reset / _enable(off) is called, so a timed region is not perturbed. */
int mlhip_timing_enable(mlhip_ctx* ctx, int on);
int mlhip_timing_reset(mlhip_ctx* ctx);
int mlhip_timing_get(mlhip_ctx* ctx, const char* name, double* avg_ms, uint64_t* launches);
/* Which kernels an EM iteration of K full-covariance components on `data` is made of (diagnostic: bench.py attributes the
 * algorithmic flops of ML/EM.cpp:190-263 to the kernels that execute them). Bits of *flags: */
#define MLHIP_PLAN_FUSED 1u          /* E-step + statistics in one kernel (small shapes, em_fused_small.hip) */
#define MLHIP_PLAN_MATRIX_ESTEP 2u   /* matrix-core E-step (em_estep_mfma4.hip), else the scalar-fed one */
#define MLHIP_PLAN_SELF_NORM 4u      /* the E-step writes log-responsibilities only; the statistics kernel normalises them
                                        (the one exponential per (sample, component) runs THERE) */
int mlhip_em_plan(const mlhip_data* data, uint32_t K, uint32_t* flags);

#ifdef __cplusplus
}
#endif
#endif /* MLHIP_H */
