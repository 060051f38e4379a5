// One EM iteration's device work in ONE kernel for small shapes (d <= 8, K <= 64): E-step + M-step statistics without the
// N x K log-responsibility block ever touching HBM -- replaces the pair EM::expectation_step / EM::maximisation_step sums
// (reference ML/EM.cpp:190-219, 221-250) when a wave can hold all K log-densities of its samples in registers.
//
// For small d the unfused pair is bandwidth-bound on exactly that block: the E-step writes N*K*8 bytes, the statistics
// kernel reads them back (d = 4, K = 16, N = 10M: 1.28 GB each way next to 0.32 GB of samples). Here a wave owns a stream
// of 64-sample tiles (lane = sample while the densities are evaluated):
//   1. lw_k = log pi_k - sum log L_jj - |W_k (x - mu_k)|^2 / 2 for all K components (same arithmetic as em_estep.hip), the
//      component records staged once per workgroup in LDS and read as broadcasts at compile-time offsets (statically
//      unrolled over K, scalar loads would need more SGPRs than exist), the K values kept in VGPRs;
//   2. m = max_k lw_k, e_k = exp(lw_k - m), s = sum_k e_k, lse = m + log s (written: log-likelihood, later
//      responsibilities), r_k = e_k / s -- one exp per (sample, component) instead of one in each of the two kernels;
//   3. r and x~ = [x - shift; 1] go to the wave's private LDS tiles and the statistics GEMM stats[K x F] += R^T Phi runs
//      on the matrix cores exactly as in em_mstats_small.hip.
// HBM traffic per iteration: X once, LSE once. The log-responsibility block is produced on demand (labels /
// responsibilities after the fit) by the ordinary E-step kernel from the same parameter records.
#include "em_close_body.hpp"
#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"

namespace mlhip {
namespace mstats {
namespace {

typedef __attribute__((address_space(3))) const double lds_cdouble;
template <int D> constexpr int xss() { return D <= 4 ? 7 : 11; }   // LDS row stride of the sample tile (d + 2 doubles used), odd
constexpr int RSS = 17;   // LDS row stride of one 16-component responsibility block, odd

/// TAIL: the workgroup that finishes LAST (a ticket per workgroup) also reduces the partial blocks and closes the iteration
/// (FusedTail, device.hpp) -- tiny fits, where the three dependent launches of an iteration cost more than their kernels.
template <int D, int RBW, int CB, bool TAIL>
__global__ __launch_bounds__(256, (D <= 4 && RBW <= 2) ? 3 : 2) void em_fused_small_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ shift,
    const double* __restrict__ params, int K, int F, double* __restrict__ lse_out, double* __restrict__ partials, int KP,
    int FP, double* __restrict__ ll_partials, FusedTail tail)
{
    constexpr int PS = D + D * (D + 1) / 2 + 1;    // estep_param_stride(D)
    constexpr int KMAX = 16 * RBW;
    constexpr int XSS = xss<D>();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* Xw = smem + (size_t)wave * (TS * XSS + TS * RSS);
    double* Rw = Xw + TS * XSS;
    double* recs = smem + 4 * (TS * XSS + TS * RSS);    // [KMAX][PS]: all component records, staged once per workgroup;
    const int da = d + 1;                               // those beyond K are neutral (zeros, coef = -inf: log-density -inf)
    for (int e = tid; e < KMAX * PS; e += 256)
        recs[e] = e < K * PS ? params[e] : (e % PS == PS - 1 ? -__builtin_inf() : 0.0);
    __syncthreads();

    int offa[CB], offb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) feature_pair(c * 16 + (lane & 15), F, da, offa[c], offb[c]);

    d4 acc[RBW][CB];
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[r][c] = d4{0.0, 0.0, 0.0, 0.0};

    const uint32_t n_tiles = (n + TS - 1) / TS;
    const uint32_t stride = gridDim.x * 4;
    const double* xbase = Xw + 16 * (lane >> 4) * XSS;
    const double* rbase = Rw + 16 * (lane >> 4) * RSS + (lane & 15);
    double ll_acc = 0.0;

    for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        // Loop-invariant values the compiler would otherwise keep in (spilled) SGPRs -- K (the guards below become scalar
        // compares) and a constant LDS address per record read; the record base lives in ONE VGPR instead, every read is
        // `ds_read base offset:imm` (see em_diag.hip: 191 SGPR spills at K = 64 before).
        int Kt = K;
        asm volatile("" : "+s"(Kt));
        lds_cdouble* recv = (lds_cdouble*)recs;
        asm volatile("" : "+v"(recv));
        const uint32_t i = tile * TS + lane;            // < n_pad: inside the allocation
        const bool live = i < n;
        double x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = xt[(size_t)j * ldx + i];

        // ---- 1. log-densities of all components (statically unrolled: the values stay in registers)
        double lwv[KMAX];
        double m = -__builtin_inf();
        // Guards are per group of 4 components (wave-uniform branches; per-component guards cost more in register copies
        // at the joins than the arithmetic they save): inside a live group a component beyond K reads a neutral record
        // (coef = -inf), which the normalisation below turns into an exact 0.
#pragma unroll
        for (int k4 = 0; k4 < KMAX; k4 += 4) {
            if (k4 < Kt) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k4 + u;
                    lds_cdouble* p = recv + k * PS;                              // LDS broadcast reads
                    double z[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) z[j] = x[j] - p[j];
                    lds_cdouble* w = p + D;
                    double q = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double y = w[j * (j + 1) / 2] * z[0];
#pragma unroll
                        for (int l = 1; l <= j; ++l) y = __builtin_fma(w[j * (j + 1) / 2 + l], z[l], y);
                        q = __builtin_fma(y, y, q);
                    }
                    const double lw = __builtin_fma(-0.5, q, p[PS - 1]);
                    lwv[k] = lw;
                    m = lw > m ? lw : m;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) lwv[k4 + u] = -__builtin_inf();
            }
        }
        // ---- 2. normalisation
        double s = 0.0;
#pragma unroll
        for (int k4 = 0; k4 < KMAX; k4 += 4) {
            if (k4 < Kt) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double e = exp_nonpos(lwv[k4 + u] - m);       // exp(-inf) = 0 for the clamped tail
                    lwv[k4 + u] = e;
                    s += e;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) lwv[k4 + u] = 0.0;
            }
        }
        const double lse = m + log(s);
        lse_out[i] = lse;
        if (live) ll_acc += lse;
        const double inv = live ? 1.0 / s : 0.0;         // padding samples contribute nothing

        // ---- 3. tiles -> LDS, statistics on the matrix cores (see em_mstats_small.hip)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < d) Xw[lane * XSS + j] = x[j] - shift[j];
        Xw[lane * XSS + d] = 1.0;
        Xw[lane * XSS + da] = 0.0;
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 16; ++it) Rw[lane * RSS + it] = lwv[rb * 16 + it] * inv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (rb * 16 < Kt) {                          // wave-uniform: skip all-zero row blocks
                __builtin_amdgcn_s_setprio(kMatrixPhasePriority);   // see em_estep_mfma4.hip
#pragma unroll 4
                for (int sg = 0; sg < TS / 4; ++sg) {
                    const double av = rbase[sg * RSS];
                    const double* xr = xbase + sg * XSS;
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const double bv = xr[offa[c]] * xr[offb[c]];
                        acc[rb][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[rb][c], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }

    // ---- epilogue: fold the 4 waves' accumulators and log-likelihood sums in fixed order
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    for (int w = 0; w < 4; ++w) {
        if (w == wave) {
#pragma unroll
            for (int r = 0; r < RBW; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int k = r * 16 + (lane >> 4) + 4 * g;
                        double* p = out + (size_t)k * FP + c * 16 + (lane & 15);
                        *p = (w == 0 ? 0.0 : *p) + acc[r][c][g];
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ll_acc += __shfl_down(ll_acc, off, 64);
    if (lane == 0) red[wave] = ll_acc;
    __syncthreads();
    if (tid == 0) ll_partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];

    if constexpr (TAIL) {
        __shared__ unsigned s_ticket;
        __threadfence();                                    // this thread's partial-block / log-likelihood writes: device-wide
        __syncthreads();
        if (tid == 0) s_ticket = atomicAdd(tail.counter, 1u);
        __syncthreads();
        if (s_ticket != gridDim.x - 1) return;              // (workgroup-uniform)
        __threadfence();                                    // the other workgroups' partial blocks
        // ---- reduction, in the order of em_reduce_kernel (em_mstats.hip): 8 outputs per pass, each summed by 32 threads over
        // the block slices b = s, s + 32, ... (ascending), the 32 slice sums added in ascending s -- the same bits
        const int total = K * F, nb = (int)gridDim.x;
        double* buf = smem;                                 // (the tiles are done with)
        for (int e0 = 0; e0 < total; e0 += 8) {
            const int o = tid & 7, sl = tid >> 3, e = e0 + o;
            double s = 0.0;
            if (e < total) {
                const int k = e / F, f = e - k * F;
                const double* p = partials + (size_t)k * FP + f;
                for (int b = sl; b < nb; b += 32) s += p[(size_t)b * KP * FP];
            }
            buf[tid] = s;
            __syncthreads();
            if (sl == 0 && e < total) {
                double t = buf[o];
#pragma unroll
                for (int q = 1; q < 32; ++q) t += buf[q * 8 + o];
                tail.stats[e] = t;
            }
            __syncthreads();
        }
        {   // log-likelihood partials: the fixed-order tree of em_reduce_kernel's extra block
            double s = 0.0;
            for (int b = tid; b < nb; b += 256) s += ll_partials[b];
            buf[tid] = s;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (tid < off) buf[tid] += buf[tid + off];
                __syncthreads();
            }
            if (tid == 0) { tail.stats[total] = buf[0]; *tail.counter = 0u; }   // (the counter is ready for the next launch)
        }
        __threadfence();                                    // the statistics are read back from memory by the closing waves
        __syncthreads();
        // ---- closing arithmetic, one wave per component, the four waves taking turns (em_close_body.hpp)
        double* sm = smem + (size_t)wave * closing::scratch_doubles(d);
        for (int k = wave; k < K; k += 4)
            closing::close_component<0, D>(tail.stats, K, d, D, shift, tail.n_global, tail.refine_limit, tail.mixing, tail.means,
                                           tail.covs, tail.records, PS, tail.info, k, lane, sm);
    }
}

template <int D, int RBW, int CB, bool TAIL>
int launch_t(const FusedArgs& a, const FusedTail& t, int grid, hipStream_t stream)
{
    constexpr int PS = D + D * (D + 1) / 2 + 1;
    constexpr int XSS = xss<D>();
    const size_t smem = sizeof(double) * (4 * ((size_t)TS * XSS + (size_t)TS * RSS) + (size_t)16 * RBW * PS);
    static_assert(4 * ((size_t)TS * XSS + (size_t)TS * RSS) >= 256 && 4 * ((size_t)TS * XSS + (size_t)TS * RSS) >= 4 * closing::scratch_doubles(D),
                  "the tail reuses the tiles' LDS");
    hipLaunchKernelGGL((em_fused_small_kernel<D, RBW, CB, TAIL>), dim3(grid), dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                       a.params, a.K, stats_count(a.d), a.lse, a.partials, RBW * 16, CB * 16, a.ll_partials, t);
    return grid;
}

template <int D, int CB, bool TAIL>
int launch_d(const FusedArgs& a, const FusedTail& t, int grid, hipStream_t stream)
{
    const int RB = (a.K + 15) / 16;
    if (RB == 1) return launch_t<D, 1, CB, TAIL>(a, t, grid, stream);
    if constexpr (!TAIL) {
        if (RB == 2) return launch_t<D, 2, CB, false>(a, t, grid, stream);
        if constexpr (CB == 1) { if (RB <= 4) return launch_t<D, 4, CB, false>(a, t, grid, stream); }
    }
    return -1;
}

}  // namespace

/// Shapes the fused kernel is used for: d <= 6 with K <= 32, d <= 4 with K <= 64 (the K densities and the
/// accumulator tiles must fit the register file).
bool em_fused_supported(int d, int K)
{
    if (d < 1 || d > 6 || K < 1) return false;       // (d = 7, 8: the two-kernel path is faster)
    const int RB = (K + 15) / 16, CB = (stats_count(d) + 15) / 16;
    return RB <= 2 || (RB <= 4 && CB == 1);
}

int em_fused_partial_rows(int K) { const int RB = (K + 15) / 16; return (RB == 1 ? 1 : RB == 2 ? 2 : 4) * 16; }
int em_fused_partial_cols(int d) { return ((stats_count(d) + 15) / 16) * 16; }

namespace {
template <bool TAIL> int launch_fused(const FusedArgs& a, const FusedTail& t, int num_cus, hipStream_t stream);
}

/// Returns the number of per-workgroup partial blocks written (stats and log-likelihood alike), or < 0.
int launch_em_fused_small(const FusedArgs& a, int num_cus, hipStream_t stream) { return launch_fused<false>(a, FusedTail{}, num_cus, stream); }

int launch_em_fused_small_tail(const FusedArgs& a, const FusedTail& t, int num_cus, hipStream_t stream)
{
    if (a.K > kFusedTailMaxK || !t.counter || !t.stats) return -1;
    return launch_fused<true>(a, t, num_cus, stream);
}

namespace {
template <bool TAIL> int launch_fused(const FusedArgs& a, const FusedTail& t, int num_cus, hipStream_t stream)
{
    if (!em_fused_supported(a.d, a.K)) return -1;
    const uint32_t n_tiles = (a.n + TS - 1) / TS;
    const int RB = (a.K + 15) / 16;
    int grid = (padded_dim(a.d) <= 4 && RB <= 2 ? 3 : 2) * num_cus;   // resident workgroups per CU of the instance
    if ((uint32_t)grid * 4 > n_tiles) grid = (int)((n_tiles + 3) / 4);
    if (grid < 1) grid = 1;
    if (grid > a.n_ll_partials) grid = a.n_ll_partials;
    const size_t block = (size_t)em_fused_partial_rows(a.K) * em_fused_partial_cols(a.d);
    if ((size_t)grid * block > a.partials_capacity) grid = (int)(a.partials_capacity / block);
    if (grid < 1) return -2;
    switch (padded_dim(a.d)) {
    case 1: return launch_d<1, 1, TAIL>(a, t, grid, stream);
    case 2: return launch_d<2, 1, TAIL>(a, t, grid, stream);
    case 3: return launch_d<3, 1, TAIL>(a, t, grid, stream);
    case 4: return launch_d<4, 1, TAIL>(a, t, grid, stream);
    case 6: return launch_d<6, 2, TAIL>(a, t, grid, stream);
    case 8: return launch_d<8, 3, TAIL>(a, t, grid, stream);
    default: return -1;
    }
}
}  // namespace

}  // namespace mstats
}  // namespace mlhip
