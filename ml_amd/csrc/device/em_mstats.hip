// M-step sufficient statistics of Gaussian-mixture EM on the gfx950 fp64 matrix cores -- replaces the
// accumulation loops of EM::maximisation_step (reference ML/EM.cpp:229, 238, 245-248: means = X*R,
// sum of weights, N*K rank-1 updates through LinearAlgebra::add_a_xxT, ML/LinearAlgebra.cpp:54-73) and of
// EM::calculate_sample_covariance (ML/EM.cpp:265-272).
//
// Formulation. With xt_i = [x_i - shift ; 1] (length d+1) every statistic the M-step needs is an entry of
//     S_k = sum_i r_ik xt_i xt_i^T        (S0 = S_k[d][d], S1' = S_k[d][0..d), M2' = S_k[0..d)[0..d))
// and S_k is symmetric, so only its F = (d+1)(d+2)/2 lower-triangular entries are formed. Stacking them,
//     stats[K x F] = R^T[K x N] * Phi[N x F],   Phi_i = vech(xt_i xt_i^T)
// is ONE dense GEMM whose B operand is generated in registers from the LDS-resident sample tile
// (one multiply per element, shared by all components): v_mfma_f64_16x16x4_f64 with
//     A[i = component][k = sample] = r_ik,  B[k = sample][j = feature] = xt_i[a(j)] * xt_i[b(j)].
// That is 2*K*F flop per sample (561 features at d = 32) instead of the 2*K*d^2 of per-component
// 16x16-tiled covariance blocks, and one pass over X and over the N x K log-responsibilities.
//
// Work split: grid.x persistent workgroups stride over 64-sample tiles; grid.y splits the (row-block,
// column-block) tile space when it does not fit one workgroup's registers. A wave holds RBW x CBW
// 16x16 accumulators (8 VGPRs each). Per-workgroup partial sums go to scratch and are combined in a
// fixed order by em_reduce_kernel (no atomics: bitwise reproducible).
#include <cstdlib>

#include "em_mstats_common.hpp"

namespace mlhip {
namespace {

using namespace mstats;

/// resp[k*ldr + i] = (labels[i] == k) for i < n (one-hot responsibilities, ML/Clustering.cpp:88), or 1 if labels == null.
__global__ __launch_bounds__(256) void fill_resp_kernel(const uint32_t* __restrict__ labels, uint32_t n_pad, uint32_t n, int K,
                                                         double* __restrict__ resp, size_t ldr)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_pad; i += gridDim.x * 256u) {
        const uint32_t lab = (labels && i < n) ? labels[i] : 0xffffffffu;
        for (int k = 0; k < K; ++k) resp[(size_t)k * ldr + i] = (i < n && (!labels || lab == (uint32_t)k)) ? 1.0 : 0.0;
    }
}

/// stats[k*F + f] = sum_b partials[b][k][f], stats[K*F] = sum of the ll partials. Fixed order, hence reproducible run to run
/// and identical on every rank: 8 outputs per workgroup, each summed by 32 threads over the block slices b = s, s + 32, ...
/// (ascending), the 32 slice sums added in ascending s. (One thread per output walking all partial blocks, as in round 1, is
/// latency-bound: 24 us for the 528 outputs of the diagonal configuration; 8 slices, round 2: 9.6 us with 768 partial blocks --
/// the chain of dependent loads per thread is what it costs, so round 3 cuts it to a quarter.)
constexpr int kRedOut = 8, kRedSlices = 32;
__global__ __launch_bounds__(256) void em_reduce_kernel(const double* __restrict__ partials, int n_blocks, int KP, int FP,
                                                         int K, int F, const double* __restrict__ ll_partials,
                                                         int n_ll, double* __restrict__ stats)
{
    __shared__ double red[256];
    const int total = K * F;
    if ((int)blockIdx.x * kRedOut < total) {
        const int o = threadIdx.x & (kRedOut - 1), sl = threadIdx.x / kRedOut;
        const int e = blockIdx.x * kRedOut + o;
        double s = 0.0;
        if (e < total) {
            const int k = e / F, f = e - k * F;
            const double* p = partials + (size_t)k * FP + f;
            int b = sl;
            for (; b + 3 * kRedSlices < n_blocks; b += 4 * kRedSlices) {          // 4 loads in flight, consumed in order
                const double v0 = p[(size_t)b * KP * FP], v1 = p[(size_t)(b + kRedSlices) * KP * FP];
                const double v2 = p[(size_t)(b + 2 * kRedSlices) * KP * FP], v3 = p[(size_t)(b + 3 * kRedSlices) * KP * FP];
                s += v0; s += v1; s += v2; s += v3;
            }
            for (; b < n_blocks; b += kRedSlices) s += p[(size_t)b * KP * FP];
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (sl == 0 && e < total) {
            double t = red[o];
#pragma unroll
            for (int q = 1; q < kRedSlices; ++q) t += red[q * kRedOut + o];
            stats[e] = t;
        }
    } else {
        // one extra block: log-likelihood partials (fixed-order tree)
        double s = 0.0;
        for (int b = threadIdx.x; b < n_ll; b += 256) s += ll_partials[b];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) stats[total] = red[0];
    }
}


/// After a self-normalising statistics pass: lse[i] holds max_k lw_ik, esum[i] the sum of exp(lw_ik - max). Finishes
/// lse[i] = max + log(esum) in place and leaves one log-likelihood partial per block (fixed-order tree: reproducible).
__global__ __launch_bounds__(256) void em_lse_finish_kernel(double* __restrict__ lse, const double* __restrict__ esum, uint32_t n,
                                                              uint32_t n_pad, double* __restrict__ ll_partials)
{
    __shared__ double red[256];
    double acc = 0.0;
    (void)n_pad;                                    // (only the live samples were written by the statistics kernel's tiles)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const double l = lse[i] + log(esum[i]);
        lse[i] = l;
        acc += l;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ll_partials[blockIdx.x] = red[0];
}

}  // namespace

namespace mstats {

Plan make_plan(int d, int K, int num_cus)
{
    Plan p{};
    const int F = stats_count(d);
    p.RB = (K + 15) / 16;
    p.CB = (F + 15) / 16;
#ifdef MLHIP_EXPERIMENTS
    static const bool force_narrow = [] { const char* e = std::getenv("MLHIP_MSTATS"); return e && e[0] == 'n'; }();
#else
    constexpr bool force_narrow = false;     // the workgroup-tile variant (experiments/em_mstats_narrow.hip) is not built
#endif
    if (d > kMaxDim) {
        // generic_dim.hip: one workgroup per (row of the packed triangle, component), ONE partial block [K][F]
        p.RBW = p.CBW = 1;
        p.n_rbg = p.n_cbg = 1;
        p.KP = K;
        p.FP = F;
        p.grid_x = big_dim_applies(d) ? big_dim_splits(d, K, num_cus) : 1;      // (big_dim.hip: one block per sample range)
        p.wg_per_cu = 1;
        return p;
    }
    p.wide = p.CB >= 5 && !force_narrow;
    if (p.wide) {
        // One 512-thread workgroup per CU; its 8 waves take the column blocks round-robin (wave w: w, w+8, ...), every
        // wave holds all (<= 4) row blocks of the group: RBW x CBW <= 20 accumulator tiles.
        // ceil(RB / 4) row-block groups of equal height (1..4 blocks: K = 48 is ONE group of 3 [r3] -- it used to be two groups
        // of 2, each staging the tile and forming the products again, without the self-normalising form)
        p.n_rbg = (p.RB + 3) / 4;
        p.RBW = (p.RB + p.n_rbg - 1) / p.n_rbg;
        // <= 5 blocks per wave: a single column group for d <= 32 (CB <= 36), ceil(CB / 40) groups of equal width above
        p.n_cbg = (p.CB + 39) / 40;
        p.CBW = (p.CB + 8 * p.n_cbg - 1) / (8 * p.n_cbg);
        p.KP = p.n_rbg * p.RBW * 16;
        p.FP = p.n_cbg * 8 * p.CBW * 16;
        p.grid_x = num_cus / (p.n_rbg * p.n_cbg);
        p.wg_per_cu = 1;
        // Few components (K <= 32) at d <= 32: a wave holds only 1 - 6 accumulator tiles, the contraction of a 64-sample tile is
        // shorter than the latency of the next tile's loads, and one workgroup per CU waits for memory most of the time. These
        // instantiations need <= 128 registers and <= 70 KB of LDS: TWO workgroups per CU [r3] (N = 1M, d = 16, K = 16: 202 ->
        // 160 us; N = 5M, d = 24, K = 32: 2.31 -> 1.97 ms; profiles/r03_mstats_wg2.txt). MLHIP_MSTATS_WG2=0: one.
        static const bool wg2_allowed = [] { const char* e = ab_env("MLHIP_MSTATS_WG2"); return !(e && e[0] == '0'); }();
        if (wg2_allowed && d <= kRegDim && p.n_rbg == 1 && p.n_cbg == 1 && (p.RBW == 1 || (p.RBW == 2 && p.CBW <= 3))) {
            p.wg_per_cu = 2;                                       // (three, where registers and LDS allow: no faster, slower at d >= 20)
            p.grid_x = p.wg_per_cu * num_cus;
        }
    } else if (p.CB <= 4 && !force_narrow) {
        // d <= 9: independent waves, each with all column blocks and up to 4 row blocks (em_mstats_small.hip)
        p.small = true;
        p.RBW = p.RB >= 3 ? 4 : (p.RB >= 2 ? 2 : 1);
        p.CBW = p.CB;
        p.n_rbg = (p.RB + p.RBW - 1) / p.RBW;
        p.n_cbg = 1;
        p.KP = p.n_rbg * p.RBW * 16;
        p.FP = p.CB * 16;
        p.grid_x = 2 * num_cus / p.n_rbg;
    } else {
        // Few column blocks (small d): 256-thread workgroups, 4 waves split the column blocks, two workgroups per CU.
        const int cb_per_wave = (p.CB + 3) / 4;
        p.RBW = p.RB >= 4 ? 4 : (p.RB >= 2 ? 2 : 1);
        p.CBW = cb_per_wave >= 2 ? 2 : 1;
        p.n_rbg = (p.RB + p.RBW - 1) / p.RBW;
        p.n_cbg = (p.CB + 4 * p.CBW - 1) / (4 * p.CBW);
        p.KP = p.n_rbg * p.RBW * 16;
        p.FP = p.n_cbg * 4 * p.CBW * 16;
        p.grid_x = 2 * num_cus / (p.n_rbg * p.n_cbg);
    }
    if (p.grid_x < 1) p.grid_x = 1;
    return p;
}

}  // namespace mstats

bool em_mstats_self_norm_supported(int d, int K, int num_cus)
{
    const Plan p = make_plan(d, K, num_cus);
    return p.wide && p.n_rbg == 1;      // all K components in one workgroup (K <= 16 RBW): see em_mstats_wide.hip, EXP == 2
}

size_t em_mstats_scratch_doubles(int d, int K, int num_cus)
{
    const Plan p = make_plan(d, K, num_cus);
    size_t need = (size_t)p.grid_x * p.KP * p.FP;
    if (mstats::em_diag_supported(d, K)) {    // the diagonal-covariance kernel: up to 2 workgroups per CU
        const size_t diag = (size_t)2 * num_cus * mstats::em_diag_partial_rows(K) * mstats::em_diag_partial_cols(d);
        if (diag > need) need = diag;
    }
    if (mstats::em_fused_supported(d, K)) {   // the fused small-shape kernel runs up to 3 workgroups per CU
        const size_t fused = (size_t)3 * num_cus * mstats::em_fused_partial_rows(K) * mstats::em_fused_partial_cols(d);
        if (fused > need) need = fused;
    }
    return need;
}

int launch_em_mstats(const MstatsArgs& a, int num_cus, hipStream_t stream)
{
    if (a.d > kMaxDim) return big_dim_applies(a.d) ? launch_em_mstats_big(a, num_cus, stream) : launch_em_mstats_generic(a, stream);
    const Plan p = make_plan(a.d, a.K, num_cus);
    const uint32_t n_tiles = (a.n + TS - 1) / TS;
    int grid_x = p.grid_x;
    if ((uint32_t)grid_x > n_tiles) grid_x = (int)(n_tiles ? n_tiles : 1);
    if ((size_t)grid_x * p.KP * p.FP > a.partials_capacity) return -2;
    if (a.mode == kFromLogRespSelfNorm && !(p.wide && p.n_rbg == 1)) return -3;
    if (p.wide) return launch_wide(a, p, grid_x, stream);
    if (p.small) return launch_small(a, p, grid_x, stream);
#ifdef MLHIP_EXPERIMENTS
    return mstats::launch_narrow(a, p, grid_x, stream);
#else
    return -1;
#endif
}

void launch_fill_responsibilities(const uint32_t* labels, uint32_t n, int K, double* resp, size_t ldr, hipStream_t stream)
{
    const uint32_t n_pad = padded_samples(n);
    uint32_t blocks = n_pad / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(fill_resp_kernel, dim3(blocks), dim3(256), 0, stream, labels, n_pad ? n_pad : kSampleTile, n, K, resp, ldr);
}

void launch_em_reduce(const MstatsArgs& a, int num_cus, int grid_x, hipStream_t stream)
{
    const Plan p = make_plan(a.d, a.K, num_cus);
    const int F = stats_count(a.d);
    const int total = a.K * F;
    const int red_blocks = (total + kRedOut - 1) / kRedOut + 1;
    // a self-normalising pass left (max, exp-sum) per sample: finish lse and produce the log-likelihood partials first
    const double* ll = a.ll_partials;
    int n_ll = a.n_ll_partials;
    if (a.mode == kFromLogRespSelfNorm) {
        const uint32_t n_pad = padded_samples(a.n);
        n_ll = (int)((n_pad / 256 < 1024u) ? n_pad / 256 : 1024u);
        hipLaunchKernelGGL(em_lse_finish_kernel, dim3(n_ll), dim3(256), 0, stream, a.lse_out, a.ll_out, a.n, n_pad, a.ll_scratch);
        ll = a.ll_scratch;
    }
    hipLaunchKernelGGL(em_reduce_kernel, dim3(red_blocks), dim3(256), 0, stream, a.partials, grid_x, p.KP, p.FP, a.K, F,
                       ll, n_ll, a.stats);
}

void launch_em_reduce_blocks(const double* partials, int n_partials, int KP, int FP, int K, int F, const double* ll_partials,
                             int n_ll, double* stats, hipStream_t stream)
{
    const int red_blocks = (K * F + kRedOut - 1) / kRedOut + 1;
    hipLaunchKernelGGL(em_reduce_kernel, dim3(red_blocks), dim3(256), 0, stream, partials, n_partials, KP, FP, K, F, ll_partials,
                       n_ll, stats);
}

void launch_ll_reduce(const double* ll_partials, int n_ll, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(em_reduce_kernel, dim3(1), dim3(256), 0, stream, nullptr, 0, 0, 0, 0, 0, ll_partials, n_ll, out);
}

}  // namespace mlhip
