// One EM iteration's device work in ONE kernel for small shapes (d <= 8, K <= 64): E-step + M-step statistics without the
// N x K log-responsibility block ever touching HBM -- replaces the pair EM::expectation_step / EM::maximisation_step sums
// (reference ML/EM.cpp:190-219, 221-250) when a wave can hold all K log-densities of its samples in registers.
//
// For small d the unfused pair is bandwidth-bound on exactly that block: the E-step writes N*K*8 bytes, the statistics
// kernel reads them back (d = 4, K = 16, N = 10M: 1.28 GB each way next to 0.32 GB of samples). Here a wave owns a stream
// of 64-sample tiles (lane = sample while the densities are evaluated):
//   1. lw_k = log pi_k - sum log L_jj - |W_k (x - mu_k)|^2 / 2 for all K components (same arithmetic as em_estep.hip), the
//      K values kept in VGPRs; the component records come from scalar registers row by row of W (SFEED: d = 7, 8, and d >= 3
//      with many samples) or from a copy staged once per workgroup in LDS, read as broadcasts at compile-time offsets;
//   2. m = max_k lw_k, e_k = exp(lw_k - m), s = sum_k e_k, lse = m + log s (written: log-likelihood, later
//      responsibilities), r_k = e_k / s -- one exp per (sample, component) instead of one in each of the two kernels;
//   3. r and x~ = [x - shift; 1] go to the wave's private LDS tiles and the statistics GEMM stats[K x F] += R^T Phi runs
//      on the matrix cores exactly as in em_mstats_small.hip.
// Few components in few dimensions (K F <= ~100 sums): em_fused_valu_kernel below keeps the statistics on the vector unit instead.
// HBM traffic per iteration: X once, LSE once. The log-responsibility block is produced on demand (labels /
// responsibilities after the fit) by the ordinary E-step kernel from the same parameter records.
#include "em_fused_valu_body.hpp"
#include "parts.hpp"

#ifndef SMALL_STATS_UNROLL
#define SMALL_STATS_UNROLL 16   // the 16 sample groups of a tile, all of them [r5] (d = 8, K = 32: 1.73 -> 1.69 ms)
#endif

namespace mlhip {
namespace mstats {
namespace {

typedef __attribute__((address_space(3))) const double lds_cdouble;
template <int D> constexpr int xss() { return D <= 4 ? 7 : 11; }   // LDS row stride of the sample tile (d + 2 doubles used), odd
constexpr int RSS = 17;   // LDS row stride of one 16-component responsibility block, odd

/// lw = coef - |W (x - mu)|^2 / 2 from one packed record [mean(D) | W lower triangle, row by row | coef] (em_estep.hip's
/// arithmetic, term by term); P: an LDS pointer (broadcast reads) or a wave-uniform global pointer (scalar loads, SCALAR = true:
/// a scheduling fence after EVERY row keeps the loads of later rows from being hoisted above the rows that use them -- at d = 8 a
/// record is 90 SGPRs, and with a fence every second row or fewer the compiler spills: 811 v_readlane + 572 v_writelane per 32
/// components against 134 + 128, 1.96 against 1.68 ms at N = 10M, K = 32. Two components, or two rows, side by side between the
/// fences -- more independent FMA chains per wave -- spill the same way).
template <int D, bool SCALAR, typename P> __device__ __forceinline__ double record_log_density(P p, const double (&x)[D])
{
    constexpr int PS = D + D * (D + 1) / 2 + 1;
    double z[D];
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = x[j] - p[j];
    P w = p + D;
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double y = w[j * (j + 1) / 2] * z[0];
#pragma unroll
        for (int l = 1; l <= j; ++l) y = __builtin_fma(w[j * (j + 1) / 2 + l], z[l], y);
        q = __builtin_fma(y, y, q);
        if constexpr (SCALAR) __builtin_amdgcn_sched_barrier(0);
    }
    return __builtin_fma(-0.5, q, p[PS - 1]);
}

/// SFEED: the component records come from SCALAR registers (wave-uniform s_load of the global records, as in em_estep.hip) instead of
/// broadcast reads of an LDS copy. At d = 7, 8 a record is 45 doubles: with the LDS feed the density loop of 32 components issues
/// 1 440 LDS reads per 64 samples next to 2 400 vector instructions, and the four SIMDs of a CU share ONE LDS pipe -- that is what
/// made the fused form slower than the two-kernel path there; scalar loads leave the LDS pipe to the statistics tiles.
template <int D, int RBW, int CB, bool SFEED = false>
__global__ __launch_bounds__(256, (D <= 4 && RBW <= 2) ? 3 : 2) void em_fused_small_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ shift,
    const double* __restrict__ params, int K, int F, double* __restrict__ lse_out, double* __restrict__ partials, int KP,
    int FP, double* __restrict__ ll_partials)
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
    if constexpr (!SFEED) {
        for (int e = tid; e < KMAX * PS; e += 256)
            recs[e] = e < K * PS ? params[e] : (e % PS == PS - 1 ? -__builtin_inf() : 0.0);
        __syncthreads();
    }

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
                    double lw;
                    if constexpr (SFEED) {
                        // a component beyond K inside a live group of four: the last real record again (no read past the K
                        // records), its log-density replaced by -inf (a scalar select)
                        lw = record_log_density<D, true>(params + (size_t)(k < Kt ? k : Kt - 1) * PS, x);
                        lw = k < Kt ? lw : -__builtin_inf();
                    } else {
                        lw = record_log_density<D, false>(recv + k * PS, x);     // LDS broadcast reads
                    }
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
#pragma unroll SMALL_STATS_UNROLL
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
}

/// FEW components in FEW dimensions (K F <= 64 numbers, F = (d+1)(d+2)/2: the reference's own benchmark regime, d = 2, K = 3,
/// Benchmarks/bm_EM.cpp): the statistics on the VECTOR unit. The matrix-core form above pads K to 16 rows and F to 16 columns of a
/// 16 x 16 x 4 product -- at d = 2, K = 8 five of six matrix-pipe cycles multiply padding (1 200 cycles per 64 samples for
/// 6 144 flops), behind an LDS round trip of r and x~. Here every lane keeps K F accumulators in registers over all its samples and
/// the lanes are summed ONCE at the end of the kernel: no tiles, no barriers, no matrix instructions in the loop
/// (em_fused_valu_body.hpp: the pass itself, shared with the device-resident loop of em_resident.hip). The densities come from
/// scalar registers as in em_estep.hip (the records are wave-uniform). Same partial-block layout as the kernel above: the reduction
/// and closing kernels are shared. Sums are formed per lane, then across lanes: fixed order, reproducible, equal to the matrix-core
/// form to rounding.
template <int D, int K>
__global__ __launch_bounds__(256) void em_fused_valu_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, const double* __restrict__ shift, const double* __restrict__ params,
    double* __restrict__ lse_out, double* __restrict__ partials, int KP, int FP, double* __restrict__ ll_partials)
{
    using S = ValuShape<D, K>;
    constexpr int F = S::F, VP = S::VP;
    __shared__ double fold[4][VP];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double acc[VP];                                           // acc[k F + f]
#pragma unroll
    for (int e = 0; e < VP; ++e) acc[e] = 0.0;
    double ll_acc = 0.0;
    double xn[D];                                             // the NEXT tile's sample, in flight while this one is worked on
    {
        const uint32_t n_tiles = (n + TS - 1) / TS, t0 = blockIdx.x * 4 + wave;
        valu_load_tile<D>(xt, ldx, t0 < n_tiles ? t0 : 0, lane, xn);
    }
    valu_tiles<D, K>(xt, ldx, n, shift, params, lse_out, blockIdx.x, gridDim.x, wave, lane, xn, acc, ll_acc);
    valu_fold<VP>(acc, ll_acc, wave, lane, fold, red);
    __syncthreads();
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    for (int e = tid; e < K * F; e += 256) {
        const int k = e / F, f = e - k * F;
        out[(size_t)k * FP + f] = ((fold[0][e] + fold[1][e]) + fold[2][e]) + fold[3][e];
    }
    if (tid == 0) ll_partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

template <int D, int RBW, int CB, bool SFEED = false>
int launch_t(const FusedArgs& a, int grid, hipStream_t stream)
{
    constexpr int PS = D + D * (D + 1) / 2 + 1;
    constexpr int XSS = xss<D>();
    const size_t smem = sizeof(double) * (4 * ((size_t)TS * XSS + (size_t)TS * RSS) + (SFEED ? 0 : (size_t)16 * RBW * PS));
    hipLaunchKernelGGL((em_fused_small_kernel<D, RBW, CB, SFEED>), dim3(grid), dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                       a.params, a.K, stats_count(a.d), a.lse, a.partials, RBW * 16, CB * 16, a.ll_partials);
    return grid;
}

/// Which feed the density loop takes its records from (tools/sfeed_sweep.sh, profiles/r04_fused_feed.txt). Scalar registers: d = 7, 8
/// always (no LDS-fed form is built there); d >= 3 from 2^19 samples on (N = 8.4M: d = 6, K = 32 1.32 -> 1.04 ms, d = 4, K = 16
/// 0.416 -> 0.371 ms, d = 3, K = 16 0.346 -> 0.336 ms; below, with one or two tiles per wave, the scalar-load latency of every
/// record row is exposed: 1 - 6 us slower) ; d = 1, 2: LDS (0.97 against 1.04 ms at d = 1, K = 64). MLHIP_FUSED_SFEED=0 / 1 forces
/// one feed for d <= 6 (A/B runs).
template <int D> bool scalar_feed(uint32_t n)
{
    if constexpr (D >= 8) return true;
    const char* e = std::getenv("MLHIP_FUSED_SFEED");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return D >= 3 && n >= (1u << 19);
}

template <int D, int CB>
int launch_d(const FusedArgs& a, int grid, hipStream_t stream)
{
    const int RB = (a.K + 15) / 16;
    if (scalar_feed<D>(a.n)) {
        if (RB == 1) return launch_t<D, 1, CB, true>(a, grid, stream);
        if (RB == 2) return launch_t<D, 2, CB, true>(a, grid, stream);
        if constexpr (CB == 1) { if (RB <= 4) return launch_t<D, 4, CB, true>(a, grid, stream); }
        return -1;
    }
    if constexpr (D < 8) {
        if (RB == 1) return launch_t<D, 1, CB>(a, grid, stream);
        if (RB == 2) return launch_t<D, 2, CB>(a, grid, stream);
        if constexpr (CB == 1) { if (RB <= 4) return launch_t<D, 4, CB>(a, grid, stream); }
    }
    return -1;
}

/// Workgroups the vector-unit form is launched with before the cut to what a CU's registers hold (launch_valu_k): one per four
/// tiles, at most 8 per CU, the capacity of the partial-block scratch.
[[maybe_unused]] int valu_grid_uncut(const FusedArgs& a, int num_cus)
{
    const uint32_t n_tiles = (a.n + TS - 1) / TS;
    int grid = 8 * num_cus;
    if ((uint32_t)grid * 4 > n_tiles) grid = (int)((n_tiles + 3) / 4);
    if (grid < 1) grid = 1;
    if (grid > a.n_ll_partials) grid = a.n_ll_partials;
    const size_t block = (size_t)em_fused_partial_rows(a.K) * em_fused_partial_cols(a.d);
    if ((size_t)grid * block > a.partials_capacity) grid = (int)(a.partials_capacity / block);
    return grid;
}

template <int D, int K> void launch_valu_k(const FusedArgs& a, int num_cus, int& grid, hipStream_t stream)
{
    if constexpr (K >= 1) {
        if (a.K == K) {
            if (grid > num_cus) {                                    // (every instantiation holds at least one workgroup per CU)
                static const int per_cu = [] {                       // workgroups a CU holds (registers: 2 K F accumulator words per lane)
                    int nb = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, em_fused_valu_kernel<D, K>, 256, 0) != hipSuccess || nb < 1) nb = 2;
                    return nb > 8 ? 8 : nb;
                }();
                const int full = per_cu * num_cus;
                if (grid > full) grid = full;
            }
            hipLaunchKernelGGL((em_fused_valu_kernel<D, K>), dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, a.n, a.shift, a.params, a.lse,
                               a.partials, em_fused_partial_rows(a.K), em_fused_partial_cols(a.d), a.ll_partials);
        } else {
            launch_valu_k<D, K - 1>(a, num_cus, grid, stream);
        }
    }
}

template <int D> int launch_valu(const FusedArgs& a, int num_cus, hipStream_t stream)
{
    int grid = valu_grid_uncut(a, num_cus);                          // (cut to what the registers of the instantiation allow, above)
    if (grid < 1) return -2;
    launch_valu_k<D, valu_max_k(D)>(a, num_cus, grid, stream);       // one instantiation per (d, K): the loops over components are
    return grid;                                                     // straight-line code, the records sit in scalar registers
}

/// Shapes that take the vector-unit form (tools/small_shape_sweep.sh, tools/small_shape_ab.sh; profiles/r04_small_shapes.txt):
/// K F <= 64 accumulators -- at every sample count (d = 2, K = 3: 15.0 against 17.7 us per iteration at N = 16 384, 52 against
/// 104 us at N = 4 194 304; never slower); up to valu_max_k (K F ~ 100: one or two waves per SIMD, a longer epilogue) from 2^20
/// samples on, where it still wins by 6 - 36 % (below that the matrix-core form is up to 4 us faster).
/// MLHIP_FUSED_VALU=0: the matrix-core form instead; =2: the vector-unit form at every N for all shapes it is built for (A/B runs,
/// tests; read per call).
[[maybe_unused]] bool valu_form_applies(const FusedArgs& a)
{
    const char* e = std::getenv("MLHIP_FUSED_VALU");
    if (e && e[0] == '0') return false;
    const int D = padded_dim(a.d);
    if (D != a.d || D > 6 || a.K > valu_max_k(D)) return false;
    return a.K * stats_count(a.d) <= 64 || a.n >= (1u << 20) || (e && e[0] == '2');     // (2: every shape it is built for -- tests)
}

}  // namespace

#if MLHIP_PART == 1
/// Shapes the fused kernel is used for: d <= 8 with K <= 32, d <= 4 with K <= 64 (the K densities and the
/// accumulator tiles must fit the register file).
bool em_fused_supported(int d, int K)
{
    if (d < 1 || d > 8 || K < 1) return false;       // (d = 7, 8: with the records from scalar registers, see the kernel)
    const int RB = (K + 15) / 16, CB = (stats_count(d) + 15) / 16;
    return RB <= 2 || (RB <= 4 && CB == 1);
}

int em_fused_partial_rows(int K) { const int RB = (K + 15) / 16; return (RB == 1 ? 1 : RB == 2 ? 2 : 4) * 16; }
int em_fused_partial_cols(int d) { return ((stats_count(d) + 15) / 16) * 16; }
#endif

// ---- compiled in six parts by padded dimension (parts.hpp): part 1 .. 6 = D 1, 2, 3, 4, 6, 8
constexpr int kPartDim = MLHIP_PART <= 4 ? MLHIP_PART : (MLHIP_PART == 5 ? 6 : 8);

/// Returns the number of per-workgroup partial blocks written (stats and log-likelihood alike), or < 0.
int MLHIP_PART_FN(launch_em_fused_small)(const FusedArgs& a, int num_cus, hipStream_t stream)
{
    constexpr int D = kPartDim, CB = D <= 4 ? 1 : (D == 6 ? 2 : 3);
    if (padded_dim(a.d) != D) return -1;
    if constexpr (valu_max_k(D) > 0) {
        if (valu_form_applies(a)) return launch_valu<D>(a, num_cus, stream);
    }
    const uint32_t n_tiles = (a.n + TS - 1) / TS;
    const int RB = (a.K + 15) / 16;
    int grid = (D <= 4 && RB <= 2 ? 3 : 2) * num_cus;   // resident workgroups per CU of the instance
    if ((uint32_t)grid * 4 > n_tiles) grid = (int)((n_tiles + 3) / 4);
    if (grid < 1) grid = 1;
    if (grid > a.n_ll_partials) grid = a.n_ll_partials;
    const size_t block = (size_t)em_fused_partial_rows(a.K) * em_fused_partial_cols(a.d);
    if ((size_t)grid * block > a.partials_capacity) grid = (int)(a.partials_capacity / block);
    if (grid < 1) return -2;
    return launch_d<D, CB>(a, grid, stream);
}

#if MLHIP_PART == 1
int launch_em_fused_small_part2(const FusedArgs&, int, hipStream_t);
int launch_em_fused_small_part3(const FusedArgs&, int, hipStream_t);
int launch_em_fused_small_part4(const FusedArgs&, int, hipStream_t);
int launch_em_fused_small_part5(const FusedArgs&, int, hipStream_t);
int launch_em_fused_small_part6(const FusedArgs&, int, hipStream_t);

/// The grid the vector-unit form would be launched with for these arguments when that is at most one workgroup per CU (no
/// register cut applies then); 0 when the shape takes another form or a larger grid. What the device-resident loop
/// (em_resident.hip) needs to reproduce the partial blocks of this kernel.
int em_fused_valu_small_grid(const FusedArgs& a, int num_cus)
{
    if (!em_fused_supported(a.d, a.K) || !valu_form_applies(a)) return 0;
    const int grid = valu_grid_uncut(a, num_cus);
    return grid >= 1 && grid <= num_cus ? grid : 0;
}

int launch_em_fused_small(const FusedArgs& a, int num_cus, hipStream_t stream)
{
    if (!em_fused_supported(a.d, a.K)) return -1;
    switch (padded_dim(a.d)) {
    case 1: return launch_em_fused_small_part1(a, num_cus, stream);
    case 2: return launch_em_fused_small_part2(a, num_cus, stream);
    case 3: return launch_em_fused_small_part3(a, num_cus, stream);
    case 4: return launch_em_fused_small_part4(a, num_cus, stream);
    case 6: return launch_em_fused_small_part5(a, num_cus, stream);
    case 8: return launch_em_fused_small_part6(a, num_cus, stream);
    default: return -1;
    }
}
#endif

}  // namespace mstats
}  // namespace mlhip
