// One EM iteration's device work for DIAGONAL covariances in ONE kernel (d <= 32, K <= 64): E-step + M-step statistics,
// X read once, only the per-sample log-sum-exp written. EXTENSION: the reference's ml::EM is full-covariance only
// (ML/EM.hpp:175); this is the same pair of loops -- EM::expectation_step (ML/EM.cpp:190-219) and the sums of
// EM::maximisation_step (:229-250) -- with every covariance restricted to its diagonal, as BASELINE.json configs[1]
// (N=1M, d=16, K=16) asks. Arithmetic per (sample, component):
//     z_j = x_j - mu_kj ;  q = sum_j (z_j * iv_kj) * z_j  (ascending j; iv = 1/sigma^2) ;  lw = log pi_k - sum_j log sigma_kj - q/2
// then per sample m = max_k lw, e_k = exp(lw_k - m), s = sum e_k, lse = m + log s, r_k = e_k / s, and per component
//     S0 = sum_i r,  S1'_j = sum_i r x~_j,  S2'_j = sum_i r x~_j^2,   x~ = x - shift  (2d + 1 numbers instead of (d+1)(d+2)/2).
//
// Mapping (the structure of em_fused_small.hip): a wave owns a stream of 64-sample tiles, lane = sample while the
// densities are evaluated (coordinates in VGPRs, component records [mu | iv | coef] staged once per workgroup in LDS and read
// as broadcasts); r and x~ then go through the wave's private LDS tiles into ONE GEMM on the fp64 matrix cores,
//     stats[K x 2d] += R^T[K x 64] * Phi[64 x 2d],   Phi_i = [x~_i ; x~_i^2],
// whose B operand is generated in registers as a product of two LDS reads (x~_j * 1 or x~_j * x~_j); S0 is a per-lane
// running sum folded once at the end. No atomics; per-workgroup partials are combined in fixed order by em_reduce_kernel.
// Bound: ~3d VALU + one exp per (sample, component) against 8d bytes per sample -- VALU/latency-bound at d = K = 16, not HBM.
#include <cstdlib>
#include "parts.hpp"
#include <type_traits>

#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"

#ifndef DIAG_STATS_UNROLL
#define DIAG_STATS_UNROLL 16   // the 16 sample groups of a slot, all of them [r5]: 73.8 -> 73.05 us
#endif

namespace mlhip {
namespace mstats {
namespace {

typedef __attribute__((address_space(3))) const double lds_cdouble;
constexpr int RSS = 17;                                      // LDS row stride of one 16-component responsibility block (odd)
template <int D> constexpr int xsd() { return (D + 2) | 1; }  // LDS row stride of the sample tile: d coords + [1, 0], odd

/// RBT = 16-component row blocks that exist (K <= 16 RBT), RBW = row blocks this workgroup accumulates (blockIdx.y picks the
/// group; every group evaluates all K densities -- the normalisation needs them), CB = 16-column blocks of the 2d features,
/// S = samples per lane: a wave's tile is 64 S samples. The density loop is bound by its LDS operand traffic (every
/// (component, dimension) needs mu and 1/sigma^2 as per-lane broadcast reads: 64 x 16 bytes through the CU's one LDS pipe for
/// 3 VALU operations), so two samples per lane halve it where the registers allow (measured at d = K = 16: 126 -> see DESIGN).
template <int D, int RBT, int RBW, int CB, int S>
__global__ __launch_bounds__(256, (D <= 16 && RBT <= 2) ? 2 : 1) void em_diag_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ shift,
    const double* __restrict__ params, int K, double* __restrict__ lse_out, double* __restrict__ partials, int KP, int FP,
    double* __restrict__ ll_partials, double ab_limit)
{
    constexpr int PS = 2 * D + 2;                             // diag_param_stride(D)
    constexpr int KMAX = 16 * RBT;
    constexpr int JC = D % 2 == 0 ? 2 : 1;                    // dimensions per operand batch of the density loop
    constexpr int XSS = xsd<D>();
    constexpr int TW = TS * S;                                // samples per wave tile
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* Xw = smem + (size_t)wave * (TS * XSS + TS * RSS);
    double* Rw = Xw + TS * XSS;
    double* recs = smem + 4 * (TS * XSS + TS * RSS);          // [KMAX][PS]: records beyond K are neutral (coef = -inf)
    constexpr int ONE = D, ZERO = D + 1;                      // LDS row: [x~_0 .. x~_(D-1) | 1 | 0]
    const int rb0 = blockIdx.y * RBW;                         // first row block accumulated here
    for (int e = tid; e < KMAX * PS; e += 256) recs[e] = params[e];
    __syncthreads();
    // Two-operation form of the density loop: with a = 1/sigma and b = -(mu - shift)/sigma the term ((x - mu) / sigma)^2 is
    // fma(a, x~, b)^2 on the shift-centred coordinate x~ -- 2 fp64 operations per (sample, component, dimension) instead of
    // 3, a third of the loop that is most of this kernel. It costs about eps (|a x~| + |b|) of every term, so it is only
    // taken while every |b| is below ab_limit (layout.hpp kDiagAbLimit: 1e-14 of a term, inside the parity tolerances);
    // otherwise -- a tight component far from the global mean -- the exact form (x - mu first) runs. Every workgroup (and
    // every rank) derives the same decision from the same records; the records in LDS are rewritten in place: mu -> b,
    // 1/sigma^2 -> a.
    bool ab;
    {
        double bm = 0.0;
        for (int e = tid; e < KMAX * D; e += 256) {
            const int k = e / D, j = e - k * D;
            if (k < K && j < d) {
                const double a = sqrt(recs[k * PS + D + j]);
                bm = fmax(bm, fabs((recs[k * PS + j] - shift[j]) * a));       // (a NaN stays out of fmax; caught below)
                if (!(a == a) || !(recs[k * PS + j] == recs[k * PS + j])) bm = __builtin_inf();
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bm = fmax(bm, __shfl_xor(bm, off, 64));
        if (lane == 0) red[wave] = bm;
        __syncthreads();
        bm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        ab = bm <= ab_limit;
        if (ab) {
            for (int e = tid; e < KMAX * D; e += 256) {
                const int k = e / D, j = e - k * D;
                const double a = sqrt(recs[k * PS + D + j]);
                const double b = -((recs[k * PS + j] - shift[j]) * a);
                recs[k * PS + j] = b;
                recs[k * PS + D + j] = a;
            }
        }
        __syncthreads();
    }

    // feature f of the GEMM: f < d -> x~_f * 1 ; d <= f < 2d -> x~_(f-d)^2 ; beyond -> 0 * 0
    int offa[CB], offb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int f = c * 16 + (lane & 15);
        offa[c] = f < d ? f : (f < 2 * d ? f - d : ZERO);
        offb[c] = f < d ? ONE : (f < 2 * d ? f - d : ZERO);
    }

    d4 acc[RBW][CB];
    double s0[RBW];                                           // lane (g, c): partial S0 of component c of each row block
#pragma unroll
    for (int r = 0; r < RBW; ++r) {
        s0[r] = 0.0;
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[r][c] = d4{0.0, 0.0, 0.0, 0.0};
    }

    const uint32_t n_tiles = (n + TW - 1) / TW;
    const uint32_t stride = gridDim.x * 4;
    const double* xbase = Xw + 16 * (lane >> 4) * XSS;
    const double* rbase = Rw + 16 * (lane >> 4) * RSS + (lane & 15);
    double ll_acc = 0.0;

    for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        // Loop-invariant values the compiler would otherwise keep in (spilled) SGPRs: K (the guards below become scalar
        // compares) and the record base, which lives in ONE VGPR so that every operand read is `ds_read base offset:imm`
        // (a constant LDS address per read gets materialised in an SGPR each: measured 279 SGPR spills at d = K = 16).
        asm volatile("" ::: "memory");
        int Kt = K;
        asm volatile("" : "+s"(Kt));
        lds_cdouble* recv = (lds_cdouble*)recs;
        asm volatile("" : "+v"(recv));
        const uint32_t i0 = tile * TW + lane;                 // sample of slot s: i0 + 64 s (< n_pad: n_pad is a multiple of 256)
        double x[S][D];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < D; ++j) x[s][j] = xt[(size_t)j * ldx + i0 + TS * s];
        if (ab) {                                             // the two-operation form works on x~ = x - shift throughout
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double sh = shift[j];
#pragma unroll
                for (int s = 0; s < S; ++s) x[s][j] -= sh;
            }
        }

        // ---- 1. log-densities of all K components (statically unrolled; wave-uniform guards per group of 4)
        double lwv[S][KMAX];
        double m[S];
#pragma unroll
        for (int s = 0; s < S; ++s) m[s] = -__builtin_inf();
        auto densities = [&](auto two_op_form) {
        constexpr bool AB = decltype(two_op_form)::value;
#pragma unroll
        for (int k4 = 0; k4 < KMAX; k4 += 4) {
            if (k4 < Kt) {
                // 4 components x JC dimensions per batch: the 2 * 4 * JC operands are read from LDS in one go, then consumed
                // by all S samples of the lane. Every q accumulates in ascending j.
                lds_cdouble* p = recv + k4 * PS;
                double q[S][4];
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[s][u] = 0.0;
#pragma unroll
                for (int j0 = 0; j0 < D; j0 += JC) {
                    double mu[4][JC], iv[4][JC];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int jj = 0; jj < JC; ++jj) {
                            mu[u][jj] = p[u * PS + j0 + jj];
                            iv[u][jj] = p[u * PS + D + j0 + jj];
                        }
#pragma unroll
                    for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                if constexpr (AB) {
                                    const double t = __builtin_fma(iv[u][jj], x[s][j0 + jj], mu[u][jj]);   // (a, b) in the (iv, mu) slots
                                    q[s][u] = __builtin_fma(t, t, q[s][u]);
                                } else {
                                    const double z = x[s][j0 + jj] - mu[u][jj];
                                    q[s][u] = __builtin_fma(z * iv[u][jj], z, q[s][u]);
                                }
                            }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double coef = p[u * PS + 2 * D];                          // records k >= K: coef = -inf
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const double lw = __builtin_fma(-0.5, q[s][u], coef);
                        lwv[s][k4 + u] = lw;
                        m[s] = lw > m[s] ? lw : m[s];
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int u = 0; u < 4; ++u) lwv[s][k4 + u] = -__builtin_inf();
            }
        }
        };
        if (ab) densities(std::true_type{}); else densities(std::false_type{});
        // ---- 2. normalisation: one exp per (sample, component)
        double inv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            double sum = 0.0;
#pragma unroll
            for (int k4 = 0; k4 < KMAX; k4 += 4) {
                if (k4 < Kt) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double e = exp_nonpos(lwv[s][k4 + u] - m[s]);       // exp(-inf) = 0 for the neutral tail
                        lwv[s][k4 + u] = e;
                        sum += e;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) lwv[s][k4 + u] = 0.0;
                }
            }
            const double lse = m[s] + log(sum);
            const bool live = i0 + TS * s < n;
            if (blockIdx.y == 0) {
                lse_out[i0 + TS * s] = lse;
                if (live) ll_acc += lse;
            }
            inv[s] = live ? 1.0 / sum : 0.0;                     // padding samples contribute nothing
        }

        // ---- 3. per 64-sample slot: tiles -> LDS, statistics on the matrix cores
#pragma unroll
        for (int s = 0; s < S; ++s) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < D; ++j) Xw[lane * XSS + j] = ab ? x[s][j] : x[s][j] - shift[j];   // shift is zero-padded to D entries
            Xw[lane * XSS + ONE] = 1.0;
            Xw[lane * XSS + ZERO] = 0.0;
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    // (rb0 is uniform over the workgroup; the row-block index is resolved at compile time per group)
                    double r = 0.0;
#pragma unroll
                    for (int g = 0; g < RBT / RBW; ++g)
                        if (rb0 == g * RBW) r = lwv[s][(g * RBW + rb) * 16 + it];
                    Rw[lane * RSS + it] = r * inv[s];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if ((rb0 + rb) * 16 < Kt) {                          // wave-uniform: skip all-zero row blocks
                    __builtin_amdgcn_s_setprio(kMatrixPhasePriority);  // see em_estep_mfma4.hip
#pragma unroll 4
                    for (int sg = 0; sg < TS / 4; ++sg) {
                        const double av = rbase[sg * RSS];           // r of (sample 16 g + sg, component lane & 15)
                        const double* xr = xbase + sg * XSS;
                        s0[rb] += av;
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
    }

    // ---- epilogue: fold the 4 waves' accumulators, S0 sums and log-likelihood sums in fixed order
    // partial block of this workgroup column: [KP][FP], row = component, columns [0, 2d) features, column 2d = S0
#pragma unroll
    for (int r = 0; r < RBW; ++r) {
        double v = s0[r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        s0[r] = v;                                               // every lane (g, c): S0 of component c over the wave's samples
    }
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    for (int w = 0; w < 4; ++w) {
        if (w == wave) {
#pragma unroll
            for (int r = 0; r < RBW; ++r) {
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int k = (rb0 + r) * 16 + (lane >> 4) + 4 * g;
                        const int f = c * 16 + (lane & 15);
                        if (f < 2 * d) {
                            double* p = out + (size_t)k * FP + f;
                            *p = (w == 0 ? 0.0 : *p) + acc[r][c][g];
                        }
                    }
                if (lane < 16) {
                    double* p = out + (size_t)((rb0 + r) * 16 + lane) * FP + 2 * d;
                    *p = (w == 0 ? 0.0 : *p) + s0[r];
                }
            }
        }
        __syncthreads();
    }
    if (blockIdx.y == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ll_acc += __shfl_down(ll_acc, off, 64);
        if (lane == 0) red[wave] = ll_acc;
        __syncthreads();
        if (tid == 0) ll_partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

/// Lanes l, l ^ 16, l ^ 32, l ^ 48 (the four lanes that hold one sample's components in the matrix cores' output layout) combine
/// their values without the LDS pipe: v_permlane16_swap / v_permlane32_swap hand every lane its partner's value.
template <bool BIT5> __device__ __forceinline__ void lane_partners(double v, double& a, double& b)
{
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (BIT5) {
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        a = __hiloint2double((int)h[0], (int)l[0]);
        b = __hiloint2double((int)h[1], (int)l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        a = __hiloint2double((int)h[0], (int)l[0]);
        b = __hiloint2double((int)h[1], (int)l[1]);
    }
}
#ifdef MLHIP_EXPERIMENTS   // (superseded by em_diag_mixed_kernel below; `make EXPERIMENTS=1` keeps it for A/B runs, MLHIP_DIAG_MIXED=0 selects it)
[[maybe_unused]] __device__ __forceinline__ double quad_max(double v)
{
    double a, b;
    lane_partners<false>(v, a, b);
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(a), "v"(b));   // (a NaN operand is dropped, as in the scalar-fed loop)
    lane_partners<true>(v, a, b);
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(a), "v"(b));
    return v;
}
[[maybe_unused]] __device__ __forceinline__ double quad_sum(double v)
{
    double a, b;
    lane_partners<false>(v, a, b);
    v = a + b;
    lane_partners<true>(v, a, b);
    return a + b;
}

/// The same iteration for K <= 16 with the component records fed from SCALAR registers instead of LDS: the
/// records are wave-uniform, so the compiler fetches them with s_load and every v_add / v_mul / v_fma of the density loop
/// takes its record operand from an SGPR pair -- one scalar source per instruction, which is what the constant bus of gfx950
/// allows; that is the reference's three-operation form  z = x - mu ; q += (z / sigma^2) z  (the two-operation form
/// fma(a, x~, b)^2 would need two). No operand goes through the LDS pipe, so ONE sample per lane costs nothing extra: half
/// the registers, FOUR waves per SIMD instead of two (the kernel is latency-bound: 3.8 tiles per wave, each starting with
/// an HBM round trip), and the statistics tiles pass through LDS in two 32-sample halves (9.2 KB per wave: 16 waves per CU).
///
/// GEMM = the log-densities on the MATRIX CORES (round 4): with a = 1 / sigma, b = -(mu - shift) / sigma the log-density is linear in
/// the features the statistics GEMM already uses, Phi = [x~^2 ; x~]:
///     lw_k = (coef_k - B2_k / 2) + sum_j (-a_kj^2 / 2) x~_j^2 + (-a_kj b_kj) x~_j  ,
/// one 16 x 2D by 2D x 16 product per 16 samples on v_mfma_f64_16x16x4 with the parameter matrix as the A operand -- (2D + 3) / 4
/// doubles per lane, loaded ONCE per kernel -- the constant as the accumulator initialiser, and the B operand read (and squared)
/// from the very sample tile the statistics phase stages in LDS. The 3 d (or 2 d) vector instructions per (sample, component) of
/// the vector forms, and with them every per-component operand feed (scalar loads, LDS broadcasts), are gone; what the vector unit
/// still does per (sample, component) is the one exponential. The output sits in the matrix cores' layout -- lane (g, c) holds
/// components g, g + 4, g + 8, g + 12 of sample c -- so maximum and sum over the components are three in-lane operations and two
/// lane exchanges (v_permlane16/32_swap), and the responsibilities go to the statistics tile from there. No coordinates stay in
/// registers across the component loop. The price is the expanded form's cancellation, ~eps 4 B2 in a log-responsibility:
/// taken only while every B2_k <= kDiagExpandLimit (layout.hpp; 4.5e-13) and the shift is the data's; the exact scalar-fed form
/// otherwise (same kernel, decided from the records, the same on every workgroup and rank).
template <int D, int CB, bool GEMM>
__global__ __launch_bounds__(256, D <= 8 ? 4 : 3) void em_diag_sgpr_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ shift,
    const double* __restrict__ params, int K, double* __restrict__ lse_out, double* __restrict__ partials, int KP, int FP,
    double* __restrict__ ll_partials, double expand_limit)
{
    constexpr int PS = 2 * D + 2;                             // diag_param_stride(D)
    constexpr int KMAX = 16;
    constexpr int XSS = xsd<D>();
    constexpr int HT = TS / 2;                                // samples per statistics half tile
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* Xw = smem + (size_t)wave * (HT * XSS + HT * RSS);
    double* Rw = Xw + HT * XSS;
    constexpr int ONE = D, ZERO = D + 1;                      // LDS row: [x~_0 .. x~_(D-1) | 1 | 0]

    int offa[CB], offb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int f = c * 16 + (lane & 15);
        offa[c] = f < d ? f : (f < 2 * d ? f - d : ZERO);
        offb[c] = f < d ? ONE : (f < 2 * d ? f - d : ZERO);
    }
    d4 acc[CB];
    double s0 = 0.0;                                          // lane (g, c): partial S0 of component c
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};

    const uint32_t n_tiles = (n + TS - 1) / TS;
    const uint32_t stride = gridDim.x * 4;
    const double* xbase = Xw + 8 * (lane >> 4) * XSS;         // half tile: lane group g reads samples 8 g + sg
    // the wave's tiles sit at fixed LDS addresses: the lane's operand pointers are loop invariants, every read in the statistics
    // loop is `ds_read base offset:imm` (no address arithmetic in a kernel whose vector instructions are its bound)
    lds_cdouble* pa[CB];
    lds_cdouble* pb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        pa[c] = (lds_cdouble*)(xbase + offa[c]);
        pb[c] = (lds_cdouble*)(xbase + offb[c]);
    }
    lds_cdouble* rbase = (lds_cdouble*)(Rw + 8 * (lane >> 4) * RSS + (lane & 15));
    double ll_acc = 0.0;

    bool expanded = false;
    if constexpr (GEMM) {
        expanded = true;                                      // every record must allow the form (wave-uniform: scalar loads; a NaN fails)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) expanded = expanded && params[(size_t)k * PS + 2 * D + 1] <= expand_limit;
    }
    if (GEMM && expanded) {
        constexpr int NQ = (2 * D + 3) / 4;                   // feature quads of Phi = [x~^2 (D) ; x~ (D)]
        const int g = lane >> 4, c = lane & 15;
        double* MS = smem + 4 * (HT * XSS + HT * RSS) + (size_t)wave * 2 * TS;     // per sample: max and exp-sum, for the tile's lse
        // A operand, lane (g, c): Theta[component c][feature 4 q + g]; accumulator initialiser, element e: component g + 4 e
        const double* __restrict__ aT = params + (size_t)KMAX * PS;
        const double* __restrict__ bT = aT + (size_t)D * KMAX;
        double th[NQ];
        int col[NQ];                                          // LDS column of the lane's feature; squared for f < D
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = 4 * q + g;
            const int j = f < D ? f : f - D;
            const double a = f < 2 * D ? aT[(size_t)j * KMAX + c] : 0.0, b = f < 2 * D ? bT[(size_t)j * KMAX + c] : 0.0;
            th[q] = f < D ? -0.5 * (a * a) : -(a * b);
            col[q] = f < 2 * D ? j : ZERO;
        }
        d4 cst;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double* __restrict__ p = params + (size_t)(g + 4 * e) * PS;
            cst[e] = p[2 * D] - 0.5 * p[2 * D + 1];
        }
        lds_cdouble* xrow = (lds_cdouble*)(Xw + c * XSS);     // the lane's sample row in 16-sample group 0 of the half tile
        for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
            const uint32_t i = tile * TS + lane;              // < n_pad (a multiple of 256)
            double x[D];
#pragma unroll
            for (int j = 0; j < D; ++j) x[j] = xt[(size_t)j * ldx + i];
#pragma unroll
            for (int j = 0; j < D; ++j) x[j] -= shift[j];     // shift is zero-padded to D entries
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __builtin_amdgcn_wave_barrier();
                if ((lane >> 5) == h) {
                    double* xw = Xw + (lane & 31) * XSS;
#pragma unroll
                    for (int j = 0; j < D; ++j) xw[j] = x[j];
                    xw[ONE] = 1.0;
                    xw[ZERO] = 0.0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- log-densities of the half's two 16-sample groups on the matrix cores, normalised in the output layout
#pragma unroll
                for (int gh = 0; gh < 2; ++gh) {
                    d4 lw = cst;
                    __builtin_amdgcn_s_setprio(kMatrixPhasePriority);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        double v = xrow[gh * 16 * XSS + col[q]];
                        if constexpr (D % 4 == 0) { if (4 * q < D) v *= v; }
                        else v = (4 * q + g < D) ? v * v : v;
                        lw = __builtin_amdgcn_mfma_f64_16x16x4f64(th[q], v, lw, 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    double m = lw[0];
                    asm("v_max_f64 %0, %0, %1" : "+v"(m) : "v"(lw[1]));
                    asm("v_max_f64 %0, %0, %1" : "+v"(m) : "v"(lw[2]));
                    asm("v_max_f64 %0, %0, %1" : "+v"(m) : "v"(lw[3]));
                    m = quad_max(m);
                    double e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = lw[u] - m;
                    exp_nonpos_n<4>(e);                       // exp(-inf) = 0 for the neutral components
                    const double sum = quad_sum(((e[0] + e[1]) + e[2]) + e[3]);
                    const uint32_t sample = tile * TS + h * HT + gh * 16 + c;
                    const double inv = sample < n ? 1.0 / sum : 0.0;               // padding samples contribute nothing
                    if (g == 0) { MS[h * HT + gh * 16 + c] = m; MS[TS + h * HT + gh * 16 + c] = sum; }
                    double* rw = Rw + (gh * 16 + c) * RSS + g;
#pragma unroll
                    for (int u = 0; u < 4; ++u) rw[4 * u] = e[u] * inv;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- statistics of the half on the matrix cores (as below)
                __builtin_amdgcn_s_setprio(kMatrixPhasePriority);
#pragma unroll
                for (int sg = 0; sg < HT / 4; ++sg) {
                    const double av = rbase[sg * RSS];
                    s0 += av;
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        const double bv = pa[cb][sg * XSS] * pb[cb][sg * XSS];
                        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[cb], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            // the tile's log-sum-exp, one sample per lane (the maxima / sums were left in LDS by the lanes g == 0)
            const double lse = MS[lane] + log(MS[TS + lane]);
            lse_out[i] = lse;
            if (i < n) ll_acc += lse;
        }
    } else
    for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        const uint32_t i = tile * TS + lane;                  // < n_pad (a multiple of 256)
        double x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = xt[(size_t)j * ldx + i];

        // ---- 1. log-densities: records from scalar registers (params + k PS is wave-uniform)
        double lwv[KMAX];
        double m = -__builtin_inf();
#pragma unroll
        for (int k4 = 0; k4 < KMAX; k4 += 4) {
            if (k4 < K) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* __restrict__ p = params + (size_t)(k4 + u) * PS;      // records k >= K: coef = -inf
                    double q = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const double z = x[j] - p[j];
                        q = __builtin_fma(z * p[D + j], z, q);
                    }
                    const double lw = __builtin_fma(-0.5, q, p[2 * D]);
                    lwv[k4 + u] = lw;
                    asm("v_max_f64 %0, %0, %1" : "+v"(m) : "v"(lw));     // m = max(m, lw); a NaN lw leaves m (as `lw > m ? lw : m` did): one
                                                                        // instruction instead of a compare and two selects
                    __builtin_amdgcn_sched_barrier(0);        // one record in scalar registers at a time
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) lwv[k4 + u] = -__builtin_inf();
            }
        }
        // ---- 2. normalisation: one exp per (sample, component)
        double sum = 0.0;
#pragma unroll
        for (int k4 = 0; k4 < KMAX; k4 += 4) {
            if (k4 < K) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double e = exp_nonpos(lwv[k4 + u] - m);                   // exp(-inf) = 0 for the neutral tail
                    lwv[k4 + u] = e;
                    sum += e;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) lwv[k4 + u] = 0.0;
            }
        }
        const double lse = m + log(sum);
        const bool live = i < n;
        lse_out[i] = lse;
        if (live) ll_acc += lse;
        const double inv = live ? 1.0 / sum : 0.0;                                  // padding samples contribute nothing
        // (formed once, in place: the two half-tile passes below run under complementary lane masks but ISSUE for the whole wave)
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] -= shift[j];                               // shift is zero-padded to D entries
#pragma unroll
        for (int it = 0; it < 16; ++it) lwv[it] *= inv;

        // ---- 3. statistics on the matrix cores, the tile in two 32-sample halves through LDS
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_wave_barrier();
            if ((lane >> 5) == h) {
                double* xw = Xw + (lane & 31) * XSS;
#pragma unroll
                for (int j = 0; j < D; ++j) xw[j] = x[j];
                xw[ONE] = 1.0;
                xw[ZERO] = 0.0;
                double* rw = Rw + (lane & 31) * RSS;
#pragma unroll
                for (int it = 0; it < 16; ++it) rw[it] = lwv[it];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_setprio(kMatrixPhasePriority);
#pragma unroll
            for (int sg = 0; sg < HT / 4; ++sg) {
                const double av = rbase[sg * RSS];               // r of (sample 8 g + sg of the half, component lane & 15)
                s0 += av;
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const double bv = pa[c][sg * XSS] * pb[c][sg * XSS];
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // ---- epilogue: fold the 4 waves' accumulators, S0 sums and log-likelihood sums in fixed order
    s0 += __shfl_xor(s0, 16, 64);
    s0 += __shfl_xor(s0, 32, 64);
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    for (int w = 0; w < 4; ++w) {
        if (w == wave) {
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = (lane >> 4) + 4 * g;
                    const int f = c * 16 + (lane & 15);
                    if (f < 2 * d) {
                        double* p = out + (size_t)k * FP + f;
                        *p = (w == 0 ? 0.0 : *p) + acc[c][g];
                    }
                }
            if (lane < 16) {
                double* p = out + (size_t)lane * FP + 2 * d;
                *p = (w == 0 ? 0.0 : *p) + s0;
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
#endif  // MLHIP_EXPERIMENTS

/// K <= 16, d <= 16 (BASELINE.json configs[1]): the two-operation density form
///     t = fma(a, x~, b) ;  q = fma(t, t, q)        a = 1 / sigma, b = -(mu - shift) / sigma
/// -- 2 vector instructions per (sample, component, dimension) instead of the 3 of the exact form -- run DIMENSION-outer: for one j the
/// operand pairs of all 16 components are contiguous (the records' trailer, layout.hpp: aT / bT, interleaved in LDS) and feed 16
/// INDEPENDENT accumulation chains per sample (q_k, which then become lw_k, e_k, r_k in place); S = 2 samples per lane share every
/// operand read. Both operands come from LDS: feeding `a` from scalar registers (one scalar source per instruction is what the
/// constant bus allows) was built in three forms and measured SLOWER than the exact kernel in spite of 20 - 24 % fewer
/// vector instructions -- scalar loads and LDS reads share one counter, so every wait in the loop becomes a wait for everything
/// (profiles/r04_diag_feed.txt). Taken while every record's flag allows it (every |b| <= kDiagAbLimit:
/// decided where the records are built, host/em_math.cpp build_diag_params and em_close_diag_kernel) and the shift is the data's;
/// otherwise the exact three-operation form runs from scalar registers. Statistics phase: that of em_diag_kernel, one row block.
template <int D, int CB, int S>
__global__ __launch_bounds__(256, 2) void em_diag_mixed_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ shift,
    const double* __restrict__ params, int K, double* __restrict__ lse_out, double* __restrict__ partials, int KP, int FP,
    double* __restrict__ ll_partials, int allow_two_op)
{
    constexpr int PS = diag_param_stride_c(D);
    constexpr int KMAX = 16;
    constexpr int XSS = xsd<D>();
    constexpr int TW = TS * S;                                // samples per wave tile
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* Xw = smem + (size_t)wave * (TS * XSS + TS * RSS);
    double* Rw = Xw + TS * XSS;
    double* bl = smem + 4 * (TS * XSS + TS * RSS);            // [D][KMAX] pairs (a, b): the records' trailer, interleaved
    constexpr int ONE = D, ZERO = D + 1;                      // LDS row: [x~_0 .. x~_(D-1) | 1 | 0]
    const double* __restrict__ aT = params + (size_t)KMAX * PS;
    const uint32_t n_tiles = (n + TW - 1) / TW;
    const uint32_t stride = gridDim.x * 4;
    // The samples of the wave's FIRST tile are requested before anything else, so the HBM round trip runs under the staging of the
    // records; every later tile loads its samples at its head, the other wave of the SIMD covering for the round trip. (Round 5 also
    // measured the next tile's samples requested slot by slot during the statistics phase: 2 - 3 us SLOWER -- the 32 coordinate
    // registers stay live through the matrix phase and the loads queue behind its LDS traffic; profiles/r05_diag_skeleton.txt.
    // The timing build keeps that placement as bit 128.)
    double x[S][D];
    {
        const uint32_t t0 = blockIdx.x * 4 + wave;
        const uint32_t i0 = (t0 < n_tiles ? t0 : 0) * TW + lane;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < D; ++j) x[s][j] = xt[(size_t)j * ldx + i0 + TS * s];
    }
    // Timing builds (`make EXPERIMENTS=1`, MLHIP_DIAG_KO=mask, tools/diag_knockout.py; profiles/r05_diag_skeleton.txt): phases of the
    // kernel switched off at run time by wave-uniform branches -- the results are WRONG by construction. 1: density loop, 2: exponentials,
    // 4: statistics phase, 8: sample loads after a wave's first tile, 16: lse store, 32: prologue, 64: partial-block flush, 128: the next
    // tile's samples requested during the statistics phase instead of at the head of the tile. The default build has no such branches.
#ifdef MLHIP_EXPERIMENTS
    const int ko = __builtin_amdgcn_readfirstlane(allow_two_op >> 8);
#define MLHIP_KO(bit) ((ko & (bit)) != 0)
#else
#define MLHIP_KO(bit) false
#endif
    if (!MLHIP_KO(32))
        for (int e = tid; e < KMAX * D; e += 256) { bl[2 * e] = aT[e]; bl[2 * e + 1] = aT[KMAX * D + e]; }
    // every record must allow the two-operation form (wave-uniform: scalar loads and compares)
    bool ab = (allow_two_op & 1) != 0;
    if (!MLHIP_KO(32)) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) ab = ab && params[(size_t)k * PS + 2 * D + 1] <= kDiagAbLimit * kDiagAbLimit;   // (B2: layout.hpp; a NaN fails)
    }
    __syncthreads();

    int offa[CB], offb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int f = c * 16 + (lane & 15);
        offa[c] = f < d ? f : (f < 2 * d ? f - d : ZERO);
        offb[c] = f < d ? ONE : (f < 2 * d ? f - d : ZERO);
    }
    d4 acc[CB];
    double s0 = 0.0;                                          // lane (g, c): partial S0 of component c
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};

    const double* xbase = Xw + 16 * (lane >> 4) * XSS;
    lds_cdouble* pa[CB];
    lds_cdouble* pb[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        pa[c] = (lds_cdouble*)(xbase + offa[c]);
        pb[c] = (lds_cdouble*)(xbase + offb[c]);
    }
    lds_cdouble* rbase = (lds_cdouble*)(Rw + 16 * (lane >> 4) * RSS + (lane & 15));
    double ll_acc = 0.0;

    for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        asm volatile("" ::: "memory");
        typedef double d2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const d2 lds_cd2;
        lds_cd2* blv = (lds_cd2*)bl;                          // ONE base register: every b read is `ds_read_b128 base offset:imm`
        asm volatile("" : "+v"(blv));
        const uint32_t i0 = tile * TW + lane;                 // sample of slot s: i0 + 64 s (< n_pad: a multiple of 256)
        if (!MLHIP_KO(128) && !MLHIP_KO(8) && tile != blockIdx.x * 4 + wave) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int j = 0; j < D; ++j) x[s][j] = xt[(size_t)j * ldx + i0 + TS * s];
        }

        // ---- 1. log-densities of all 16 component slots (records k >= K are neutral: a = b = 0, coef = -inf)
        double lwv[S][KMAX];
        double m[S];
#pragma unroll
        for (int s = 0; s < S; ++s) m[s] = -__builtin_inf();
        if (ab) {
#pragma unroll
            for (int j = 0; j < D; ++j) {                     // the two-operation form works on x~ = x - shift throughout
                const double sh = shift[j];
#pragma unroll
                for (int s = 0; s < S; ++s) x[s][j] -= sh;
            }
            // (no zeroing of the 32 sums: the first dimension's term is a plain product -- fma(t, t, +0) and t * t are the same bits --,
            // 32 register moves per 64 samples less [r5]; the timing build zeroes them when the density loop is knocked out)
            if (MLHIP_KO(1)) {
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) lwv[s][k] = 0.0;
            }
            // Software pipeline over the dimensions, in HALVES of 8 components: the operand pairs (a, b) of a half -- eight 16-byte LDS
            // broadcast reads -- are re-loaded for dimension j + 1 as soon as dimension j's arithmetic on that half has been ISSUED,
            // and land while the other half's 16 S instructions run (no second operand buffer; the waits are the compiler's own
            // counted lgkmcnt). Within a half the first operation of four components is issued before their second: a dependent
            // v_fma_f64 right behind its producer stalls.
            constexpr int KH = KMAX / 2;
                d2 pv[2][KH];
            if (!MLHIP_KO(1)) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < KH; ++c) pv[h][c] = blv[h * KH + c];
#pragma unroll
                for (int j = 0; j < D; ++j) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int g = 0; g < KH; g += 4) {
                            double t[4][S];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int s = 0; s < S; ++s) t[u][s] = __builtin_fma(pv[h][g + u][0], x[s][j], pv[h][g + u][1]);
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int s = 0; s < S; ++s)
                                    lwv[s][h * KH + g + u] = j == 0 ? t[u][s] * t[u][s] : __builtin_fma(t[u][s], t[u][s], lwv[s][h * KH + g + u]);
                        }
                        if (j + 1 < D) {
#pragma unroll
                            for (int c = 0; c < KH; ++c) pv[h][c] = blv[(j + 1) * KMAX + h * KH + c];
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const double coef = params[(size_t)k * PS + 2 * D];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const double lw = __builtin_fma(-0.5, lwv[s][k], coef);
                    lwv[s][k] = lw;
                    asm("v_max_f64 %0, %0, %1" : "+v"(m[s]) : "v"(lw));             // (a NaN lw leaves m, as `lw > m ? lw : m` does)
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    const double* __restrict__ p = params + (size_t)k * PS;
                    double q[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) q[s] = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j)
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const double z = x[s][j] - p[j];
                            q[s] = __builtin_fma(z * p[D + j], z, q[s]);
                        }
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const double lw = __builtin_fma(-0.5, q[s], p[2 * D]);
                        lwv[s][k] = lw;
                        asm("v_max_f64 %0, %0, %1" : "+v"(m[s]) : "v"(lw));
                    }
                    __builtin_amdgcn_sched_barrier(0);        // one record in scalar registers at a time
                } else {
#pragma unroll
                    for (int s = 0; s < S; ++s) lwv[s][k] = -__builtin_inf();
                }
            }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double sh = shift[j];                   // shift is zero-padded to D entries
#pragma unroll
                for (int s = 0; s < S; ++s) x[s][j] -= sh;
            }
        }
        // ---- 2. normalisation: one exp per (sample, component)
        double inv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            double sum = 0.0;
#pragma unroll
            for (int k8 = 0; k8 < KMAX; k8 += 8) {                                // eight chains side by side (exp_nonpos.hpp)
                double e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = lwv[s][k8 + u] - m[s];
                if (!MLHIP_KO(2)) exp_nonpos_n<8>(e);                            // exp(-inf) = 0 for the neutral tail
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    lwv[s][k8 + u] = e[u];
                    sum += e[u];
                }
            }
            const double lse = m[s] + log(sum);
            const bool live = i0 + TS * s < n;
            if (!MLHIP_KO(16)) lse_out[i0 + TS * s] = lse;
            if (live) ll_acc += lse;
            inv[s] = live ? 1.0 / sum : 0.0;                     // padding samples contribute nothing
        }

        // ---- 3. per 64-sample slot: tiles -> LDS, statistics on the matrix cores
        if (!MLHIP_KO(4))
#pragma unroll
        for (int s = 0; s < S; ++s) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < D; ++j) Xw[lane * XSS + j] = x[s][j];
            Xw[lane * XSS + ONE] = 1.0;
            Xw[lane * XSS + ZERO] = 0.0;
#pragma unroll
            for (int it = 0; it < 16; ++it) Rw[lane * RSS + it] = lwv[s][it] * inv[s];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifdef MLHIP_EXPERIMENTS
            if (tile + stride < n_tiles && !MLHIP_KO(8) && MLHIP_KO(128)) {   // (timing builds) slot s of the NEXT tile: in flight during the matrix phase
                const uint32_t inext = (tile + stride) * TW + lane;
#pragma unroll
                for (int j = 0; j < D; ++j) x[s][j] = xt[(size_t)j * ldx + inext + TS * s];
            }
#endif
            __builtin_amdgcn_s_setprio(kMatrixPhasePriority);
#pragma unroll DIAG_STATS_UNROLL
            for (int sg = 0; sg < TS / 4; ++sg) {
                const double av = rbase[sg * RSS];               // r of (sample 16 g + sg, component lane & 15)
                s0 += av;
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const double bv = pa[c][sg * XSS] * pb[c][sg * XSS];
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // ---- epilogue: the 4 waves' accumulators and S0 sums folded in fixed order  (((0 + w0) + w1) + w2) + w3  -- through LDS (every
    // wave parks its values in its own, now idle, tile region; one barrier; one store per output). Round 4 chained the waves through
    // global memory, a read-modify-write and a barrier per wave: 4.5 us of a 77 us kernel.
    s0 += __shfl_xor(s0, 16, 64);
    s0 += __shfl_xor(s0, 32, 64);
    constexpr int WREG = TS * XSS + TS * RSS;                  // doubles per wave region (>= 64 (4 CB + 1))
    static_assert(64 * (4 * CB + 1) <= WREG, "the parked accumulators must fit a wave's tile region");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) Xw[(c * 4 + g) * 64 + lane] = acc[c][g];
    Xw[4 * CB * 64 + lane] = s0;
    __syncthreads();
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    if (!MLHIP_KO(64)) {
        for (int e = tid; e < 4 * CB * 64; e += 256) {
            const int l = e & 63, g = (e >> 6) & 3, c = e >> 8;
            const int k = (l >> 4) + 4 * g, f = c * 16 + (l & 15);
            if (f < 2 * d)
                out[(size_t)k * FP + f] = (((0.0 + smem[e]) + smem[WREG + e]) + smem[2 * WREG + e]) + smem[3 * WREG + e];
        }
        if (tid < 16) {
            const int e = 4 * CB * 64 + tid;
            out[(size_t)tid * FP + 2 * d] = (((0.0 + smem[e]) + smem[WREG + e]) + smem[2 * WREG + e]) + smem[3 * WREG + e];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ll_acc += __shfl_down(ll_acc, off, 64);
    if (lane == 0) red[wave] = ll_acc;
    __syncthreads();
    if (tid == 0) ll_partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
#undef MLHIP_KO
}

/// Shapes em_diag_mixed_kernel serves: one row block of components, d <= 16. In a `make EXPERIMENTS=1` build MLHIP_DIAG_MIXED=0 /
/// MLHIP_DIAG_GEMM=1 select em_diag_sgpr_kernel (the exact scalar-fed form / its matrix-core density path) and MLHIP_DIAG_SGPR=0 the
/// LDS-fed general kernel instead (A/B runs).
[[maybe_unused]] inline bool diag_sgpr_applies(int d, int K)
{
#ifdef MLHIP_EXPERIMENTS
    const char* e = std::getenv("MLHIP_DIAG_SGPR");
    if (e && e[0] == '0') return false;
#endif
    return K <= 16 && padded_dim(d) <= 16;
}
[[maybe_unused]] inline bool diag_mixed_applies(int d, int K)
{
#ifdef MLHIP_EXPERIMENTS
    const char* e = std::getenv("MLHIP_DIAG_MIXED");
    const char* g = std::getenv("MLHIP_DIAG_GEMM");
    if ((e && e[0] == '0') || (g && g[0] == '1')) return false;
#endif
    return K <= 16 && padded_dim(d) <= 16;
}

/// Timing builds only (em_diag_mixed_kernel): MLHIP_DIAG_KO=mask in a `make EXPERIMENTS=1` build; 0 otherwise.
[[maybe_unused]] inline int diag_knockouts()
{
#ifdef MLHIP_EXPERIMENTS
    const char* e = std::getenv("MLHIP_DIAG_KO");
    return e && *e ? (std::atoi(e) << 8) : 0;
#else
    return 0;
#endif
}

constexpr int rbw_of(int RBT) { return RBT >= 2 ? 2 : 1; }

/// MLHIP_DIAG_AB=0: the exact form of the density loop always (A/B runs).
inline double diag_ab_limit()
{
    const char* e = std::getenv("MLHIP_DIAG_AB");          // (read per launch: tests switch it inside one process)
    return e && e[0] == '0' ? -1.0 : kDiagAbLimit;
}

#ifdef MLHIP_EXPERIMENTS
/// Largest B2 = sum_j ((mu_j - shift_j) / sigma_j)^2 of a component for which the expanded form runs on the matrix cores
/// (layout.hpp kDiagExpandLimit; MLHIP_DIAG_EXPAND_LIMIT overrides it -- tests, error measurements).
[[maybe_unused]] inline double diag_expand_limit()
{
    const char* e = std::getenv("MLHIP_DIAG_EXPAND_LIMIT");   // (read per launch: tests switch it inside one process)
    return e && *e ? std::atof(e) : kDiagExpandLimit;
}
#endif

/// Samples per lane: 2 while coordinates + densities of both fit the registers of 2 waves per SIMD, else 1.
constexpr int samples_per_lane(int D, int RBT) { return (D <= 16 && RBT == 1) || (D <= 8 && RBT == 2) ? 2 : 1; }

template <int D, int RBT>
int launch_t(const DiagArgs& a, int grid, hipStream_t stream)
{
    constexpr int CB = (2 * D + 15) / 16, RBW = rbw_of(RBT), PS = 2 * D + 2, XSS = xsd<D>(), S = samples_per_lane(D, RBT);
    if constexpr (RBT == 1 && D <= 16) {
        if (diag_sgpr_applies(a.d, a.K) && diag_mixed_applies(a.d, a.K)) {
            const size_t smem = sizeof(double) * (4 * ((size_t)TS * XSS + (size_t)TS * RSS) + (size_t)32 * D);
            hipLaunchKernelGGL((em_diag_mixed_kernel<D, CB, 2>), dim3(grid), dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                               a.params, a.K, a.lse, a.partials, em_diag_partial_rows(a.K), em_diag_partial_cols(a.d), a.ll_partials,
                               ((a.two_op && diag_ab_limit() > 0) ? 1 : 0) | diag_knockouts());
            return grid;
        }
#ifdef MLHIP_EXPERIMENTS
        if (diag_sgpr_applies(a.d, a.K)) {
            const char* e = std::getenv("MLHIP_DIAG_GEMM");        // 1: log-densities on the matrix cores while the guard allows (A/B runs)
            const bool gemm = e && e[0] == '1';
            const size_t smem = sizeof(double) * (4 * ((size_t)(TS / 2) * XSS + (size_t)(TS / 2) * RSS) + (gemm ? 4 * 2 * (size_t)TS : 0));
            if (gemm)
                hipLaunchKernelGGL((em_diag_sgpr_kernel<D, CB, true>), dim3(grid), dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                                   a.params, a.K, a.lse, a.partials, em_diag_partial_rows(a.K), em_diag_partial_cols(a.d), a.ll_partials,
                                   (a.two_op && diag_ab_limit() > 0) ? diag_expand_limit() : -1.0);
            else
                hipLaunchKernelGGL((em_diag_sgpr_kernel<D, CB, false>), dim3(grid), dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                                   a.params, a.K, a.lse, a.partials, em_diag_partial_rows(a.K), em_diag_partial_cols(a.d), a.ll_partials, -1.0);
            return grid;
        }
    }
    {
#else
        return -1;                                                 // (one row block, d <= 16: always the kernel above in the default build)
    } else {
#endif
        const size_t smem = sizeof(double) * (4 * ((size_t)TS * XSS + (size_t)TS * RSS) + (size_t)16 * RBT * PS);
        hipLaunchKernelGGL((em_diag_kernel<D, RBT, RBW, CB, S>), dim3(grid, RBT / RBW), dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d,
                           a.shift, a.params, a.K, a.lse, a.partials, em_diag_partial_rows(a.K), em_diag_partial_cols(a.d),
                           a.ll_partials, diag_ab_limit());
        return grid;
    }
}

template <int D>
int launch_d(const DiagArgs& a, int grid, hipStream_t stream)
{
    const int RB = (a.K + 15) / 16;
    if (RB == 1) return launch_t<D, 1>(a, grid, stream);
    if (RB == 2) return launch_t<D, 2>(a, grid, stream);
    if (RB <= 4) return launch_t<D, 4>(a, grid, stream);
    return -1;
}

}  // namespace

// ---- compiled in six parts by padded dimension (parts.hpp): 1: D = 1, 2; 2: 3, 4; 3: 6, 8; 4: 12, 16; 5: 20, 24; 6: 28, 32
int MLHIP_PART_FN(launch_em_diag)(const DiagArgs& a, int grid, hipStream_t stream)
{
    switch (padded_dim(a.d)) {
#if MLHIP_PART == 1
    case 1: return launch_d<1>(a, grid, stream);
    case 2: return launch_d<2>(a, grid, stream);
#elif MLHIP_PART == 2
    case 3: return launch_d<3>(a, grid, stream);
    case 4: return launch_d<4>(a, grid, stream);
#elif MLHIP_PART == 3
    case 6: return launch_d<6>(a, grid, stream);
    case 8: return launch_d<8>(a, grid, stream);
#elif MLHIP_PART == 4
    case 12: return launch_d<12>(a, grid, stream);
    case 16: return launch_d<16>(a, grid, stream);
#elif MLHIP_PART == 5
    case 20: return launch_d<20>(a, grid, stream);
    case 24: return launch_d<24>(a, grid, stream);
#elif MLHIP_PART == 6
    case 28: return launch_d<28>(a, grid, stream);
    case 32: return launch_d<32>(a, grid, stream);
#endif
    default: return -1;
    }
}

#if MLHIP_PART == 1
int launch_em_diag_part2(const DiagArgs&, int, hipStream_t);
int launch_em_diag_part3(const DiagArgs&, int, hipStream_t);
int launch_em_diag_part4(const DiagArgs&, int, hipStream_t);
int launch_em_diag_part5(const DiagArgs&, int, hipStream_t);
int launch_em_diag_part6(const DiagArgs&, int, hipStream_t);

bool em_diag_supported(int d, int K) { return d >= 1 && d <= kRegDim && K >= 1 && K <= 64; }
int em_diag_partial_rows(int K) { const int RB = (K + 15) / 16; return (RB == 1 ? 1 : RB == 2 ? 2 : 4) * 16; }
int em_diag_partial_cols(int d) { return (2 * d + 1 + 15) / 16 * 16; }

/// Workgroups in x the launch will use for (d, K, n) -- also the number of partial blocks / log-likelihood partials.
int em_diag_grid(int d, int K, uint32_t n, int num_cus)
{
    const int RB = (K + 15) / 16;
    const int D = padded_dim(d), RBT = RB == 1 ? 1 : (RB == 2 ? 2 : 4);
    const bool mixed = diag_sgpr_applies(d, K) && diag_mixed_applies(d, K);
    const bool sgpr = diag_sgpr_applies(d, K) && !mixed;
    const uint32_t tw = (uint32_t)TS * (sgpr ? 1 : (mixed ? 2 : samples_per_lane(D, RBT)));
    const uint32_t n_tiles = (n + tw - 1) / tw;
    const int groups = RB >= 4 ? 2 : 1;                          // row-block groups in grid.y
    int per_cu = sgpr ? (D <= 8 ? 4 : 3) : ((D <= 16 && RB <= 2) ? 2 : 1);   // workgroups the registers / LDS admit per CU
    int grid = per_cu * num_cus / groups;
    if ((uint32_t)grid * 4 > n_tiles) grid = (int)((n_tiles + 3) / 4);
    return grid < 1 ? 1 : grid;
}

int launch_em_diag(const DiagArgs& a, int num_cus, hipStream_t stream)
{
    if (!em_diag_supported(a.d, a.K)) return -1;
    int grid = em_diag_grid(a.d, a.K, a.n, num_cus);
    if (grid > a.n_ll_partials) grid = a.n_ll_partials;
    const size_t block = (size_t)em_diag_partial_rows(a.K) * em_diag_partial_cols(a.d);
    if ((size_t)grid * block > a.partials_capacity) grid = (int)(a.partials_capacity / block);
    if (grid < 1) return -2;
    const int D = padded_dim(a.d);
    if (D <= 2) return launch_em_diag_part1(a, grid, stream);
    if (D <= 4) return launch_em_diag_part2(a, grid, stream);
    if (D <= 8) return launch_em_diag_part3(a, grid, stream);
    if (D <= 16) return launch_em_diag_part4(a, grid, stream);
    if (D <= 24) return launch_em_diag_part5(a, grid, stream);
    return launch_em_diag_part6(a, grid, stream);
}
#endif

}  // namespace mstats
}  // namespace mlhip
