// Internal interface between the runtime (runtime.cpp) and the gfx950 kernels (*.hip).
// Everything here is launched on the caller's hipStream_t; no function synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "layout.hpp"

namespace mlhip {

// ---- data movement --------------------------------------------------------------------------------
/// dst[j*ldd + i0 + i] = src[i*lds + j] for i < n, j < d (sample-major -> dimension-major).
void launch_transpose_to_dim_major(const double* src, int64_t lds, int d, uint64_t n, double* dst, size_t ldd,
                                   uint64_t i0, hipStream_t stream);
/// sums[j] = sum_i xt[j*ldx + i] (deterministic two-stage reduction); scratch >= d * 1024 doubles.
/// maxabs[j] = max_i |xt[j*ldx + i]|; scratch >= d * 1024 doubles.
void launch_column_maxabs(const double* xt, size_t ldx, int d, uint64_t n, double* scratch, double* maxabs, hipStream_t stream);
void launch_column_sums(const double* xt, size_t ldx, int d, uint64_t n, double* scratch, double* sums, hipStream_t stream);

/// RandomPartition::init's running means on the resident block (data_kernels.hip): means[k*d + j] and sizes[k] are continued
/// over the rows order[offsets[k] .. offsets[k+1]) of cluster k, in that order. All pointers device memory.
void launch_random_partition(const double* xt, size_t ldx, int d, int K, const uint32_t* order, const uint32_t* offsets,
                             double* means, double* sizes, hipStream_t stream);

/// One K-means++ draw on the device (data_kernels.hip), in two launches around the ranks' exchange of their weight sums:
/// update: weights = first ? dist : min(weights, dist) over this rank's n rows, block sums / offsets, out[0] = this rank's sum,
///         out[1] = out[2] = default_index (the last row of the whole sample);
/// find:   the bounds [lo, hi] (global row indices, out[1] / out[2], as doubles) of the row std::discrete_distribution would return
///         for the canonical uniform u, given the sum of the ranks before this one (offset) and of all ranks (total).
/// bsum / boff: kpp_blocks(n) doubles of scratch each.
int kpp_blocks(uint32_t n);
void launch_kpp_update(double* weights, const double* dist, uint32_t n, int first, double default_index, double* bsum, double* boff,
                       double* out, hipStream_t stream);
void launch_kpp_find(const double* weights, uint32_t n, const double* bsum, const double* boff, double offset, double total, double u,
                     double delta, uint64_t row0, uint64_t n_global, double* out, hipStream_t stream);

/// Fixed-order sum of the n <= kGroupMaxShards buffers slots.p[0..n) (`count` doubles each) into out: the all-reduce of a device
/// group whose shards share a process (runtime/group.cpp).
constexpr int kGroupMaxShards = 64;
struct GroupSumSlots { const double* p[kGroupMaxShards]; };
void launch_group_sum(const GroupSumSlots& slots, int n, double* out, size_t count, hipStream_t stream);

// ---- EM ----------------------------------------------------------------------------------------------
struct EstepArgs {
    const double* xt; size_t ldx; uint32_t n; int D;      // D = padded dimension
    const double* params; int K;                            // K records of estep_param_stride(D)
    double* lw; size_t ldr;                                 // out: unnormalised log-responsibilities [K][ldr]
    double* lse;                                            // out: log sum_k exp(lw) per sample
    double* ll_partials; int n_ll_partials;                 // out: per-block sums of lse (grid size)
    // em_estep_mfma4 only:
    const double* shift;                                    // device, D doubles (zero-padded): the fold's centre
    int fold;                                               // records carry -W (mu - shift) instead of the mean (d <= 32)
    int with_lse;                                           // 0: write lw only (the statistics kernel normalises)
    int num_cus;                                            // compute units of the context's device (0: ask the current device)
    double* scratch; size_t scratch_doubles;                // a block free for the launch (the statistics partials): the big tier's q parts
};
/// Returns the grid size used (= number of ll partials written), or <0 if D is not instantiated.
int launch_em_estep(const EstepArgs& a, hipStream_t stream);
/// d > kMaxDim (generic_dim.hip): the same passes in a plain form, any dimension.
int launch_em_estep_generic(const EstepArgs& a, hipStream_t stream);
#ifdef MLHIP_EXPERIMENTS
/// 16x16x4 block-triangular variant (experiments/em_estep_mfma16.hip); params use the estep_mfma_param_stride(D) layout.
int launch_em_estep_mfma(const EstepArgs& a, int num_cus, hipStream_t stream);
#endif
/// 4x4-block triangular variant (em_estep_mfma4.hip); params use the estep_mfma4_param_stride(D) record layout.
int launch_em_estep_mfma4(const EstepArgs& a, int num_cus, hipStream_t stream);
#ifdef MLHIP_EXPERIMENTS
/// Component-stationary variant of it (experiments/em_estep_cs.hip): FOLD form, lw only (a.fold && !a.with_lse), same records;
/// returns <0 when it does not serve the shape (em_estep_cs_supported).
bool em_estep_cs_supported(int D, int K);
int launch_em_estep_cs(const EstepArgs& a, int num_cus, hipStream_t stream);
#endif

enum MstatsMode : int {
    kFromLogResp = 0,   // r = exp(lw - lse)          (after an E-step)
    kFromResp = 1,      // r = lw                     (plain responsibilities: caller-given, one-hot, or all ones)
    kFromLogRespSelfNorm = 2   // r = e_k / sum_k e_k with the normalisation done by the statistics kernel itself (wide kernel, one
                               // row-block group); launch_em_reduce then also finishes lse = max + log(sum) and the ll partials
};
struct MstatsArgs {
    const double* xt; size_t ldx; uint32_t n; int d;        // d = true dimension
    const double* shift;                                     // device, d doubles
    const double* lw; size_t ldr; const double* lse; int K; int mode;   // lw: [K][ldr], ldr >= n_pad
    double* partials; size_t partials_capacity;              // scratch (doubles)
    const double* ll_partials; int n_ll_partials;            // summed into stats[K*F] (may be null/0)
    double* stats;                                           // out: device, K*F + 1 doubles
    double* ll_scratch;                                      // kFromLogRespSelfNorm: >= 1024 doubles for the ll partials
    double* lse_out; double* ll_out;                         // kFromLogRespSelfNorm: per-sample max (-> lse) and exp-sum (n_pad each)
};
int launch_em_mstats_generic(const MstatsArgs& a, hipStream_t stream);   // d > kMaxDim: writes ONE partial block [K][F]
/// 128 < d <= 1024 on the matrix cores (big_dim.hip); the plain tier above it and with MLHIP_BIG_DIM=0.
bool big_dim_applies(int d);
bool big_dim_kmeans_applies(int d);                                       // (its K-means kernel: any d > 128)
int big_dim_splits(int d, int K, int num_cus);
int launch_em_estep_big(const EstepArgs& a, int num_cus, hipStream_t stream);
int launch_em_mstats_big(const MstatsArgs& a, int num_cus, hipStream_t stream);   // writes big_dim_splits partial blocks [K][F]
/// Whether the statistics kernel chosen for (d, K) can normalise log-responsibilities itself (mode kFromLogRespSelfNorm).
bool em_mstats_self_norm_supported(int d, int K, int num_cus);
/// Fused E-step + statistics for small shapes (em_fused_small.hip): params are the estep_param_stride(D) records.
struct FusedArgs {
    const double* xt; size_t ldx; uint32_t n; int d;
    const double* shift; const double* params; int K;
    double* lse;                                             // out: per-sample log-sum-exp
    double* partials; size_t partials_capacity;              // scratch: [grid][KP][FP]
    double* ll_partials; int n_ll_partials;                  // out: per-workgroup log-likelihood sums (grid of them)
};
namespace mstats {
bool em_fused_supported(int d, int K);
int em_fused_partial_rows(int K);
int em_fused_partial_cols(int d);
int launch_em_fused_small(const FusedArgs& a, int num_cus, hipStream_t stream);
/// The grid launch_em_fused_small would use for these arguments if the shape takes the vector-unit form with at most one
/// workgroup per CU; 0 otherwise.
int em_fused_valu_small_grid(const FusedArgs& a, int num_cus);
}
/// The whole loop of EM::fit (ML/EM.cpp:143-170) in ONE launch for short fits (em_resident.hip): resident workgroups, one
/// hand-off of the partial statistics per iteration, the closing arithmetic and the convergence test on the device. Pointers in
/// ring order: iteration i reads the records of slot i % 3 (slot 0 on entry) and leaves parameters / records in slot (i + 1) % 3,
/// exactly what the launches of runtime/em_loop.cpp leave.
struct ResidentArgs {
    const double* xt; size_t ldx; uint32_t n; int d; int K;
    const double* shift;
    double* records[3];                                      // estep_param_stride(d) records (device)
    double* info[3]; double* mixing[3]; double* means[3]; double* covs[3];   // the packs' parts (device, or pinned host memory)
    double* xch;                                             // device: em_resident_exchange_doubles(d, K, vgrid) doubles
    unsigned* sync;                                          // device: [arrivals, give-up flag], zeroed before the launch
    int vgrid;                                               // partial blocks = workgroups (em_fused_valu_small_grid)
    double n_global, refine_limit, atol, rtol, ll_offset;    // log-likelihood = sum / n_global - ll_offset
    uint32_t max_steps;
    double* history;                                         // pinned host: max_steps log-likelihoods
    uint32_t* result;                                        // pinned host: [status, iterations evaluated, converged]; status 1 = loop over,
                                                             // 2 = iteration `result[1]` flagged a refinement (not closed), 3 = a wait gave up
    unsigned long long* profile;                             // null, or kResidentStamps clock stamps per iteration (MLHIP_RESIDENT_PROFILE=1)
};
namespace mstats {
constexpr int kResidentSlices = 32;                          // (= em_reduce_kernel's slices: the same summation order)
constexpr int kResidentMaxGrid = 64;                          // workgroups (= partial blocks = CUs used): N <= 16 384
constexpr int kResidentStamps = 24;
size_t em_resident_exchange_doubles(int d, int K, int vgrid);
bool em_resident_supported(int d, int K, int vgrid, int num_cus);
bool launch_em_resident(const ResidentArgs& a, hipStream_t stream);
}
/// Diagonal-covariance EM iteration in one kernel (em_diag.hip): params are em_diag_partial_rows(K) records of
/// diag_param_stride(padded_dim(d)) -- K real ones, then neutral padding (coef = -inf) -- and shift holds padded_dim(d)
/// doubles (zeros beyond d).
struct DiagArgs {
    const double* xt; size_t ldx; uint32_t n; int d;
    const double* shift; const double* params; int K;
    double* lse;                                             // out: per-sample log-sum-exp
    double* partials; size_t partials_capacity;              // scratch: [grid][KP][FP]
    double* ll_partials; int n_ll_partials;                  // out: per-workgroup log-likelihood sums
    int two_op;                                              // the records' (a, b) operands were built for THIS shift (the data's): the
                                                             // mixed-feed kernel may take its two-operation density form; 0 for a
                                                             // refinement pass about another shift (exact form)
};
namespace mstats {
bool em_diag_supported(int d, int K);                        // d <= 32, K <= 64
int em_diag_partial_rows(int K);
int em_diag_partial_cols(int d);
int em_diag_grid(int d, int K, uint32_t n, int num_cus);
/// Returns the number of per-workgroup partial blocks written (stats and log-likelihood alike), or < 0.
int launch_em_diag(const DiagArgs& a, int num_cus, hipStream_t stream);
}
/// M-step closing arithmetic + next E-step records on the device (em_close.hip): one workgroup per component.
struct CloseArgs {
    const double* stats; int K; int d; int D;                // all-reduced statistics [K][F] + ll sum (F: full or diagonal)
    const double* shift; double n_global;                    // shift: D doubles, zero padded
    int layout;                                              // full covariances: 0 = estep_param_stride records, 2 = mfma4 records
    double refine_limit;                                     // <= 0: no refinement flags
    double* mixing; double* means; double* covs;             // out (device): [K], [K*d], [K*d*d] (diagonal: [K*d] variances)
    double* records;                                         // out (device): the next E-step's K records
    double* info;                                            // out (device): [ll_sum | refine flag (K) | max |W (mu - shift)| (K)]
    double* work;                                            // d > 64: em_close_work_doubles(d, K) doubles of device scratch
};
size_t em_close_info_doubles(int K);
bool em_close_supported(int d);                              // d <= 64: one wave per component, the matrices in LDS (em_close.hip);
                                                             // 64 < d <= 1024: panelled, the matrices in global memory (em_close_big.hip)
size_t em_close_work_doubles(int d, int K);                  // 0 for d <= 64
bool em_close_big_supported(int d);
size_t em_close_big_work_doubles(int d, int K);
void launch_em_close_big(const CloseArgs& a, hipStream_t stream);
double* em_close_big_param_area(double* work, int d, int K);   // K (d d + d + 1) + 2 K + 1 doubles behind the matrices
/// Records of GIVEN parameters (a fit's first E-step): a.mixing / a.means / a.covs are device inputs, a.stats unused.
void launch_em_records_big(const CloseArgs& a, hipStream_t stream);
void launch_em_close(const CloseArgs& a, hipStream_t stream);
void launch_em_close_diag(const CloseArgs& a, hipStream_t stream);
/// Fixed-order combination of `n_partials` blocks [KP][FP] (and of the log-likelihood partials) into stats[K*F (+1)].
void launch_em_reduce_blocks(const double* partials, int n_partials, int KP, int FP, int K, int F, const double* ll_partials,
                             int n_ll, double* stats, hipStream_t stream);
/// resp[k*ldr + i] = (labels[i] == k), or 1 everywhere when labels == nullptr; columns n..n_pad-1 are zeroed.
void launch_fill_responsibilities(const uint32_t* labels, uint32_t n, int K, double* resp, size_t ldr, hipStream_t stream);
/// Doubles of scratch the statistics kernel needs for (d, K).
size_t em_mstats_scratch_doubles(int d, int K, int num_cus);
/// Main statistics kernel; returns the number of per-workgroup partials written (>0) or <0 on error.
int launch_em_mstats(const MstatsArgs& a, int num_cus, hipStream_t stream);
/// Fixed-order combination of those partials (and of the log-likelihood partials) into a.stats.
void launch_em_reduce(const MstatsArgs& a, int num_cus, int n_partials, hipStream_t stream);
/// out[0] = sum of n_ll log-likelihood partials (fixed order).
void launch_ll_reduce(const double* ll_partials, int n_ll, double* out, hipStream_t stream);

struct RespArgs {
    const double* lw; size_t ldr; const double* lse; uint32_t n; int K;
    double* resp; size_t ldo;       // out (may be null): resp[k*ldo + i]
    uint32_t* labels;               // out (may be null)
};
void launch_em_responsibilities(const RespArgs& a, hipStream_t stream);

// ---- K-means -----------------------------------------------------------------------------------------
struct KmeansArgs {
    const double* xt; size_t ldx; uint32_t n; int D; int d;
    const double* centroids; int K;            // device, [K][D] (padded coordinates zero)
    const double* scale;                       // device, d doubles: per-dimension power-of-two scale of the exact sums
    uint32_t* labels; const uint32_t* old_labels; int have_old;
    double* min_dist;                          // out (may be null): per-sample min squared distance
    int accumulate;                            // also accumulate per-cluster sums / counts
    double* partials; size_t partials_capacity;
    double* cnorm;                             // device scratch, K rounded up to 16 doubles: -|c_k|^2/2 (chunked-table kernel)
    double* out;                               // device: [inertia, n_changed, counts(K), sums(K*d)]
};
size_t kmeans_scratch_doubles(int d, int K, int num_cus);
/// The whole step loop of KMeans::fit_once in one launch of one workgroup (kmeans_resident.hip): small blocks, few clusters, the
/// dimensions whose step runs on the direct-form kernel. `out` is host-visible pinned memory:
/// [steps, converged, inertia, label buffer, counts(K), centroids(K d), old centroids(K d)].
constexpr int kKmResidentMaxN = 4096, kKmResidentMaxK = 32;
struct KmResidentArgs {
    const double* xt; size_t ldx; uint32_t n; int D, d, K;
    double* cent;                              // device, [K][D]: the starting table in, the final one out
    const double* scale;                       // device, d doubles (as KmeansArgs)
    uint32_t* labels[2]; int label_buf, have_old;   // the two label buffers, which one holds the assignment before, whether it does
    double* min_dist;
    uint32_t max_steps; double atol;
    double* out;
    int copies;                                // (set by the launcher)
};
bool kmeans_resident_supported(int D, int d, int K, uint64_t n);
bool launch_kmeans_resident(const KmResidentArgs& a, hipStream_t stream);
/// Assignment kernel; returns the number of per-workgroup partials (>0) or <0 on error.
int launch_kmeans_assign(const KmeansArgs& a, int num_cus, hipStream_t stream);
void launch_kmeans_assign_generic(const KmeansArgs& a, int grid, size_t pstride, hipStream_t stream);   // d > kMaxDim (generic_dim.hip)
/// 128 < d <= 1024 (big_dim.hip): the same exact arithmetic, register-blocked; returns the partial blocks used, 0: not applicable.
int launch_kmeans_assign_big(const KmeansArgs& a, int grid_max, size_t pstride, hipStream_t stream);
void launch_kmeans_reduce(const KmeansArgs& a, int n_partials, hipStream_t stream);
/// update_step's closing arithmetic on the (all-reduced) output block [inertia, changed, counts(K), sums(K*d)]: the sums
/// become the means IN PLACE (empty cluster -> origin, ML/KMeans.cpp:184) and are written as the next centroid table
/// next[K][D] (padded coordinates zero). `mirror` (may be null): host-visible pinned memory that receives a copy of the whole block
/// from the same kernel -- the host then only waits for the stream; a separate hipMemcpyAsync of these 2 + K (d + 1) doubles goes
/// through a copy engine and costs more than the kernel itself.
void launch_kmeans_close(double* out, int K, int d, int D, double* next, double* mirror, hipStream_t stream);
/// launch_kmeans_reduce + launch_kmeans_close in one launch (no all-reduce in between). `ticket`: one unsigned of device memory, zeroed once;
/// `ticket_base`: the tickets drawn from it so far -- the return value (this launch's workgroups) is added to it by the caller.
unsigned launch_kmeans_reduce_close(const KmeansArgs& a, int n_partials, int D, double* next, double* mirror, unsigned* ticket,
                                    unsigned ticket_base, hipStream_t stream);

}  // namespace mlhip
