// M-step sufficient statistics, wide variant (>= 5 column blocks, i.e. d >= 8): see em_mstats.hip for the formulation
// (one fp64-MFMA GEMM stats[K x F] = R^T Phi with Phi generated in registers). This file holds the decomposition used
// for the headline shapes:
//
//   * one 512-thread workgroup per CU; the 64-sample tile (x~ rows and responsibilities of up to 64 components) is staged
//     ONCE per CU; the exp(lw - lse) normalisation is applied while staging, once per (sample, component);
//   * wave w takes column blocks w, w+8, w+16, ... (round-robin: at d = 32 waves 0-3 hold 5, waves 4-7 hold 4 of the 36
//     blocks, and the two waves sharing a SIMD -- w and w+4 -- together always hold 9) and ALL row blocks of the group:
//     RBW x CBW <= 4 x 5 accumulator tiles, so every generated B operand (one multiply, two LDS reads) feeds 4 MFMAs;
//   * software pipeline over tiles: global loads of tile t+1 are issued (unconditionally, clamped addresses) right after
//     the barrier and stay in flight during the MFMA phase of tile t; LDS tiles are double-buffered -> one barrier per
//     tile;
//   * responsibilities come in one of three forms (template EXP): 0 = plain r (caller-given, one-hot, all ones), 1 = log-
//     responsibilities with the per-sample log-sum-exp given (r = exp(lw - lse)), 2 = log-responsibilities only: the kernel
//     NORMALISES THEM ITSELF while staging (the E-step then needs no exp at all: one exp per (sample, component) in the whole
//     iteration instead of two). For that the staging roles are laid out so that the K values of one sample sit in ONE wave
//     (lane = sample-in-8 + 8 * component group, each thread 2 RBW consecutive components): max and sum over them are three
//     xor-shuffle steps inside the wave, no extra barrier; lanes of component group 0 write the sample's maximum and the sum of
//     its exponentials, which a small follow-up kernel (em_lse_finish_kernel, em_mstats.hip) turns into lse and the log-
//     likelihood partials. Needs all K components in this workgroup (a single row-block group, K <= 64).
#include <type_traits>
#include "parts.hpp"

#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"

#ifndef MSTATS_UNROLL_GEN
#define MSTATS_UNROLL_GEN 16   // the 16 sample groups of a tile, all of them (see the contraction loop)
#endif

namespace mlhip {
namespace mstats {
namespace {

constexpr int NW = 8;   // waves per workgroup
typedef __attribute__((address_space(3))) const double lds_cdouble;

/// Partner values for all-reductions over lane bits 3, 4, 5 without the LDS pipe (a __shfl_xor of a double is two
/// ds_bpermute: ~100 cycles of latency each, six of them in a row in the staging phase where every wave of the CU waits):
/// bit 3 by a DPP row rotation, bits 4 and 5 by v_permlane16_swap / v_permlane32_swap.
__device__ __forceinline__ double partner_xor8(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xf, 0xf, false);   // row_ror:8
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool BIT5> __device__ __forceinline__ void partners_swap(double v, double& a, double& b)
{
    // a = v, b = v; swap: afterwards (a, b) hold {own half-or-row value, partner's} such that op(a, b) is the pairwise result
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
__device__ __forceinline__ double allreduce_max_bits345(double v)
{
    double a, b;
    v = fmax(v, partner_xor8(v));
    partners_swap<false>(v, a, b);
    v = fmax(a, b);
    partners_swap<true>(v, a, b);
    return fmax(a, b);
}
__device__ __forceinline__ double allreduce_sum_bits345(double v)
{
    double a, b;
    v += partner_xor8(v);
    partners_swap<false>(v, a, b);
    v = a + b;
    partners_swap<true>(v, a, b);
    return a + b;
}

template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

/// CBT > 0 (round 3; single row-block and column-block group, d <= 32): the RBW x CBT accumulator tiles of the GEMM are dealt to
/// the 8 waves as (column block, row block) UNITS in column-major order, ceil / floor(RBW CBT / 8) each, instead of whole column
/// blocks with all their row blocks. At d = 16 (10 column blocks) two waves used to hold 2 x 4 tiles and six waves 1 x 4 -- the
/// contraction ran at the pace of the two; now every wave holds 5. The wave's loop is specialised per wave (switch on the
/// wave index), so which units share an operand product is known at compile time: one product per column block a wave touches.
template <int RBW, int CBW, int EXP, int DM, int CBT = 0>
__global__ __launch_bounds__(512, 2) void em_mstats_wide_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, int D, const double* __restrict__ shift,
    const double* __restrict__ lw, size_t ldr, const double* __restrict__ lse, int K, int n_rbg, int CB_total,
    double* __restrict__ partials, int KP, int FP, double* __restrict__ lse_out, double* __restrict__ ll_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int RS = RBW * 16 + 1;     // odd row stride of the responsibility tile
    constexpr int NXV = DM / NW;         // x rows staged per thread (4; 8 for d > 32; 16 for d > 64)
    constexpr int XS = tile_stride<DM>();
    constexpr bool DOUBLE_BUF = DM <= kMidDim;   // two (x~, R) tiles of 64 samples do not fit LDS above d = 64
    constexpr int NRV = RBW * 16 / NW;   // responsibility rows staged per thread (2 * RBW)
    const int da = d + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbg = blockIdx.y % n_rbg, cbg = blockIdx.y / n_rbg;
    const int rb0 = rbg * RBW;
    const int F = da * (da + 1) / 2;

    // column block c of this wave: cbg*8*CBW + c*8 + wave
    constexpr int NU = CBT > 0 ? (RBW * CBT + NW - 1) / NW : 1;   // balanced form: units per wave (at most)
    constexpr int NOFF = CBT > 0 ? NU : CBW;
    int offa[NOFF], offb[NOFF];
    const int u_lo = CBT > 0 ? wave * (RBW * CBT) / NW : 0, u_hi = CBT > 0 ? (wave + 1) * (RBW * CBT) / NW : 0;
    if constexpr (CBT > 0) {
        // unit u = (column block u / RBW, row block u % RBW); slot j of this wave is unit u_lo + j
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int cb = (u_lo + j) / RBW;
            int a, b;
            feature_pair(cb * 16 + (lane & 15), u_lo + j < u_hi ? F : 0, da, a, b);
            offa[j] = a;
            offb[j] = b;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
            const int cb = (cbg * CBW + c) * NW + wave;
            int a, b;
            feature_pair(cb * 16 + (lane & 15), cb < CB_total ? F : 0, da, a, b);
            offa[c] = a;
            offb[c] = b;
        }
    }
    // With CBW = ceil(CB / 8) column blocks per wave and a single column group (d <= 32), only a wave's LAST block can
    // fall outside the matrix (wave-uniform); all others are unconditional, which keeps them in one basic block. With
    // several column groups (d > 32) earlier blocks of the last group may be outside too: they multiply the zero slot.
    const bool last_active = (cbg * CBW + CBW - 1) * NW + wave < CB_total;

    constexpr int AR = CBT > 0 ? 1 : RBW, AC = CBT > 0 ? NU : CBW;
    d4 acc[AR][AC];
#pragma unroll
    for (int r = 0; r < AR; ++r)
#pragma unroll
        for (int c = 0; c < AC; ++c) acc[r][c] = d4{0.0, 0.0, 0.0, 0.0};

    // staging role: sample sS of the tile, rows wave, wave+8, ... For EXP == 2 the responsibilities have their own role:
    // sample sR = 8 wave + (lane & 7), components NRV cg + it with cg = lane >> 3 -- one wave holds all K values of its 8 samples
    // (contiguous components per thread: with the odd row stride RS the 16 lanes of an LDS write phase then hit 16 different
    // bank pairs; the interleaved assignment cg + 8 it collides up to 8-fold and cost 1 ms at the headline shape).
    const int sS = lane;
    const int sR = EXP == 2 ? 8 * wave + (lane & 7) : lane;
    const int cg = lane >> 3;
    double xv[NXV], rv[NRV], lv = 0.0;
    auto prefetch = [&](uint32_t tile) {
        const uint32_t i = tile * TS + sS;           // < n_pad: always inside the allocation
#pragma unroll
        for (int it = 0; it < NXV; ++it) xv[it] = xt[(size_t)min(wave + NW * it, D - 1) * ldx + i];
        if constexpr (EXP == 2) {
#pragma unroll
            for (int it = 0; it < NRV; ++it) rv[it] = lw[(size_t)min(cg * NRV + it, K - 1) * ldr + tile * TS + sR];
        } else {
#pragma unroll
            for (int it = 0; it < NRV; ++it) rv[it] = lw[(size_t)min(rb0 * 16 + wave + NW * it, K - 1) * ldr + i];
            if (EXP) lv = lse[i];
        }
    };
    auto stage = [&](double* Xb, double* Rb, uint32_t tile) {
        if constexpr (EXP == 2) {
            const uint32_t i = tile * TS + sR;
            const bool live = i < n;
            double m = -__builtin_inf();
#pragma unroll
            for (int it = 0; it < NRV; ++it) {
                if (cg * NRV + it >= K) rv[it] = -__builtin_inf();          // components beyond K: exp(-inf) = 0
                m = fmax(m, rv[it]);
            }
            m = allreduce_max_bits345(m);
            double sum = 0.0;
#pragma unroll
            for (int it = 0; it < NRV; ++it) {
                rv[it] = exp_nonpos(rv[it] - m);
                sum += rv[it];
                __builtin_amdgcn_sched_barrier(0);          // one exp at a time: interleaved they spill next to 160 accumulator registers
            }
            sum = allreduce_sum_bits345(sum);
            const double inv = live ? 1.0 / sum : 0.0;                      // padding samples contribute nothing
#pragma unroll
            for (int it = 0; it < NRV; ++it) Rb[sR * RS + cg * NRV + it] = rv[it] * inv;
            // lse = m + log(sum) is finished by a separate pass over these two N-vectors (em_lse_finish_kernel): a log in
            // this loop, next to 160 accumulator registers, cost 0.8 ms at the headline shape in spills and scheduling
            if (cg == 0 && blockIdx.y == 0) {
                lse_out[i] = m;
                ll_out[i] = sum;
            }
        } else {
        const bool live = tile * TS + sS < n;
#pragma unroll
        for (int it = 0; it < NRV; ++it) {
            double r = rv[it];
            if (EXP) r = exp_nonpos(r - lv);
            const bool valid = live && (rb0 * 16 + wave + NW * it < K);
            Rb[sS * RS + wave + NW * it] = valid ? r : 0.0;
        }
        }
#pragma unroll
        for (int it = 0; it < NXV; ++it) {
            const int j = wave + NW * it;
            if (j < d) Xb[sS * XS + j] = xv[it] - shift[j];   // wave-uniform index: scalar load
        }
        if (wave == 0) {
            Xb[sS * XS + d] = 1.0;
            Xb[sS * XS + da] = 0.0;
        }
    };

    const uint32_t n_tiles = (n + TS - 1) / TS;
    constexpr int tile_doubles = TS * XS + TS * RS;
    int buf = 0;
    if (blockIdx.x < n_tiles) prefetch(blockIdx.x);
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= DOUBLE_BUF ? 1 : 0) {
        if constexpr (!DOUBLE_BUF) __syncthreads();   // single tile buffer: everyone is done reading the previous tile
        double* Xb = smem + buf * tile_doubles;
        double* Rb = Xb + TS * XS;
        stage(Xb, Rb, tile);
        __syncthreads();
        const uint32_t next = tile + gridDim.x;
        prefetch(next < n_tiles ? next : tile);      // the last iteration re-reads its own tile (discarded)
        // ---- contraction: 16 groups of 4 samples. Lane group g = lane>>4 takes sample sg + 16 g: rows 16 apart are
        // 32 banks apart for both tiles (odd strides), so the two rows of a half-wave never collide.
        const double* xbase = Xb + 16 * (lane >> 4) * XS;
        const double* rbase = Rb + 16 * (lane >> 4) * RS + (lane & 15);
        __builtin_amdgcn_s_setprio(kMatrixPhasePriority);   // see em_estep_mfma4.hip (11.78 -> 11.59 ms at d = 32, K = 64)
        if constexpr (CBT > 0) {
            auto contract = [&](auto w_) {
                constexpr int W = w_, U = RBW * CBT, LO = W * U / NW, HI = (W + 1) * U / NW, NJ = HI - LO;
                auto step = [&](int sg) {
                    const double* xr = xbase + sg * XS;
                    const double* rr = rbase + sg * RS;
                    double av[RBW];
#pragma unroll
                    for (int r = 0; r < RBW; ++r) av[r] = rr[r * 16];
                    double bv = 0.0;
                    static_for<NJ>([&](auto j_) {
                        constexpr int j = j_, u = LO + j;
                        if constexpr (j == 0 || u / RBW != (u - 1) / RBW) bv = xr[offa[j]] * xr[offb[j]];   // first unit of a column block
                        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u % RBW], bv, acc[0][j], 0, 0, 0);
                    });
                };
                for (int sg = 0; sg < TS / 4; sg += 2) {     // two sample groups per trip (written out: `#pragma unroll 2` is refused
                    step(sg);                                // on some of the eight specialisations)
                    step(sg + 1);
                }
            };
            switch (wave) {
            case 0: contract(std::integral_constant<int, 0>{}); break;
            case 1: contract(std::integral_constant<int, 1>{}); break;
            case 2: contract(std::integral_constant<int, 2>{}); break;
            case 3: contract(std::integral_constant<int, 3>{}); break;
            case 4: contract(std::integral_constant<int, 4>{}); break;
            case 5: contract(std::integral_constant<int, 5>{}); break;
            case 6: contract(std::integral_constant<int, 6>{}); break;
            default: contract(std::integral_constant<int, 7>{}); break;
            }
        } else if constexpr (RBW * CBW < 20) {
        // Every shape but the headline's 4 x 5 tiles per wave (which sits at 256 registers): the lane's operand addresses (tile base +
        // coordinate offset) are formed ONCE per tile; inside the loop every LDS read is `ds_read base offset:imm` off running
        // pointers that advance by U sample groups per trip -- otherwise one v_add_u32 per read [r3] (d = 24, K = 64: 4.09 -> 3.97 ms).
        // [r5] also above d = 32, where it had been left out: N = 1M, K = 32, d = 128 11.58 -> 10.54 ms, d = 96 6.50 -> 5.85.
        constexpr int U = TS / 4;                    // [r5] the whole tile unrolled: no pointer advances at all (U = 4 until round 5: d = 128, K = 32 10.5 -> 10.1 ms,
                                                     // d = 32, K = 32 3.11 -> 3.00, d = 28, K = 48 3.53 -> 3.42, d = 12, K = 48 0.574 -> 0.553)
        lds_cdouble* pa[CBW];
        lds_cdouble* pb[CBW];
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
            pa[c] = (lds_cdouble*)(xbase + offa[c]);
            pb[c] = (lds_cdouble*)(xbase + offb[c]);
        }
        lds_cdouble* pr = (lds_cdouble*)rbase;
#pragma unroll 1
        for (int sg0 = 0; sg0 < TS / 4; sg0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double av[RBW];
#pragma unroll
                for (int r = 0; r < RBW; ++r) av[r] = pr[u * RS + r * 16];
#pragma unroll
                for (int c = 0; c < CBW - 1; ++c) {
                    const double bv = pa[c][u * XS] * pb[c][u * XS];
#pragma unroll
                    for (int r = 0; r < RBW; ++r)
                        acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
                }
                constexpr int c = CBW - 1;                   // (see the loop below for the last block)
                const double bv = pa[c][u * XS] * pb[c][u * XS];
                if (last_active) {
#pragma unroll
                    for (int r = 0; r < RBW; ++r)
                        acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < CBW; ++c) {
                pa[c] += U * XS;
                pb[c] += U * XS;
            }
            pr += U * RS;
        }
        } else {
        // [r5] unrolled over the WHOLE tile (two sample groups per trip until round 5): the operand addresses become one register per column
        // block plus immediates -- no address arithmetic between the matrix instructions --, 249 registers and no scratch instead of
        // 256 + 8 bytes: headline statistics kernel 11.66 -> 11.09 ms (42.7 - 42.8 it/s from 41.6 on the same box), d = 64, K = 64 13.14 -> 12.53.
#pragma unroll MSTATS_UNROLL_GEN
        for (int sg = 0; sg < TS / 4; ++sg) {
            const double* xr = xbase + sg * XS;
            const double* rr = rbase + sg * RS;
            double av[RBW];
#pragma unroll
            for (int r = 0; r < RBW; ++r) av[r] = rr[r * 16];
#pragma unroll
            for (int c = 0; c < CBW - 1; ++c) {
                const double bv = xr[offa[c]] * xr[offb[c]];
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
            }
            // The last block's operand is formed unconditionally (padding columns read the zero slot), so its LDS
            // reads are scheduled with the others; only its MFMAs are skipped by the waves that do not own a block.
            constexpr int c = CBW - 1;
            const double bv = xr[offa[c]] * xr[offb[c]];
            if (last_active) {
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
            }
        }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue: partials[blockIdx.x][k][f]; C/D layout of v_mfma_f64_16x16x4: col = lane&15, row = (lane>>4) + 4*reg
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    if constexpr (CBT > 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int u = u_lo + j;
            if (u < u_hi) {
                const int cb = u / RBW, rb = u % RBW;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = rb * 16 + (lane >> 4) + 4 * g;
                    out[(size_t)k * FP + cb * 16 + (lane & 15)] = acc[0][j][g];
                }
            }
        }
    } else {
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
            const int cb = (cbg * CBW + c) * NW + wave;
            if (cb < CB_total) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = (rb0 + r) * 16 + (lane >> 4) + 4 * g;
                    out[(size_t)k * FP + cb * 16 + (lane & 15)] = acc[r][c][g];
                }
            }
        }
    }
}

#ifdef MLHIP_EXPERIMENTS
// ---- 64 < d <= 128: half tiles, double-buffered (round 5) ----------------------------------------------------------------------
// Above d = 64 the kernel above cannot double-buffer its 64-sample tile (x~ rows of 131 doubles + responsibilities: 84 KB, twice that does
// not fit the CU's 160 KB): a barrier BEFORE staging, the staging of every tile -- normalisation with its exponentials, x - shift, 20 LDS
// writes per thread -- exposed, the matrix pipe 62 % busy (d = 128, K = 32: 46 TFLOP/s, the dominant kernel of that shape). Here the
// tile is cut in HALVES of 32 samples, two of which fit; the structure is the d <= 64 kernel's -- stage half h into one buffer, ONE
// barrier, request half h + 1 from memory, contract half h -- so that a wave staging the next half runs beside the SIMD's other wave
// still in its matrix phase. Same partial-block layout, same three responsibility forms (EXP), plain dealing of column blocks.
// MEASURED SLOWER (N = 1M, K = 32: d = 128 12.39 against 11.61 ms, d = 96 6.79 / 6.52, d = 72 4.10 / 4.03): twice the barriers for the
// same staging work per sample, and a wave still cannot overlap its own staging with its own matrix phase. `make EXPERIMENTS=1`
// library with MLHIP_MSTATS_HALF=1 only (A/B runs; profiles/r05_big_dim.txt).
__device__ __forceinline__ double dpp_row_ror(double v, int ctrl_is_12)
{
    const int lo = ctrl_is_12 ? __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x12C, 0xf, 0xf, false)
                              : __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x124, 0xf, 0xf, false);
    const int hi = ctrl_is_12 ? __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x12C, 0xf, 0xf, false)
                              : __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x124, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
/// The value of lane ^ 4 (row rotations by 4 and by 12 give lane + 4 and lane - 4 inside a row of 16).
__device__ __forceinline__ double partner_xor4(double v, int lane)
{
    const double up = dpp_row_ror(v, 0), down = dpp_row_ror(v, 1);
    return (lane & 4) ? down : up;
}
[[maybe_unused]] __device__ __forceinline__ double allreduce_max_bits2345(double v, int lane)
{
    double a, b;
    v = fmax(v, partner_xor4(v, lane));
    v = fmax(v, partner_xor8(v));
    partners_swap<false>(v, a, b);
    v = fmax(a, b);
    partners_swap<true>(v, a, b);
    return fmax(a, b);
}
[[maybe_unused]] __device__ __forceinline__ double allreduce_sum_bits2345(double v, int lane)
{
    double a, b;
    v += partner_xor4(v, lane);                   // (pairwise with the SAME partner on both sides: every lane of a sample ends with the same bits)
    v += partner_xor8(v);
    partners_swap<false>(v, a, b);
    v = a + b;
    partners_swap<true>(v, a, b);
    return a + b;
}

template <int RBW, int CBW, int EXP>
__global__ __launch_bounds__(512, 2) void em_mstats_half_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, int D, const double* __restrict__ shift,
    const double* __restrict__ lw, size_t ldr, const double* __restrict__ lse, int K, int n_rbg, int CB_total,
    double* __restrict__ partials, int KP, int FP, double* __restrict__ lse_out, double* __restrict__ ll_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int DM = kMaxDim, HT = 32;            // half tile: 32 samples
    constexpr int RS = RBW * 16 + 1;                // odd row stride of the responsibility tile
    constexpr int XS = tile_stride<DM>();
    constexpr int NXV = DM / 16;                    // x rows staged per thread (8)
    constexpr int NRV = RBW;                        // responsibility rows staged per thread
    const int da = d + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbg = blockIdx.y % n_rbg, cbg = blockIdx.y / n_rbg;
    const int rb0 = rbg * RBW;
    const int F = da * (da + 1) / 2;
    int offa[CBW], offb[CBW];
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
        const int cb = (cbg * CBW + c) * NW + wave;
        int a, b;
        feature_pair(cb * 16 + (lane & 15), cb < CB_total ? F : 0, da, a, b);
        offa[c] = a;
        offb[c] = b;
    }
    const bool last_active = (cbg * CBW + CBW - 1) * NW + wave < CB_total;
    d4 acc[RBW][CBW];
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[r][c] = d4{0.0, 0.0, 0.0, 0.0};

    // staging roles: sample sS of the half tile, rows rowS + 16 it (x~ rows; component rows for EXP < 2). EXP == 2: sample sR = 4 wave +
    // (lane & 3), components NRV cg + it with cg = lane >> 2 -- one wave holds all K values of its 4 samples.
    const int sS = lane & 31, rowS = 2 * wave + (lane >> 5);
    const int sR = EXP == 2 ? 4 * wave + (lane & 3) : sS;
    const int cg = lane >> 2;
    double xv[NXV], rv[NRV], lv = 0.0;
    auto prefetch = [&](uint32_t h) {
        size_t ldx_ = ldx;
        asm volatile("" : "+s"(ldx_));              // (row addresses formed per call, not carried through the matrix phase)
        const uint32_t i = h * HT + sS;             // < n_pad: always inside the allocation
#pragma unroll
        for (int it = 0; it < NXV; ++it) xv[it] = xt[(size_t)min(rowS + 16 * it, D - 1) * ldx_ + i];
        if constexpr (EXP == 2) {
#pragma unroll
            for (int it = 0; it < NRV; ++it) rv[it] = lw[(size_t)min(cg * NRV + it, K - 1) * ldr + h * HT + sR];
        } else {
#pragma unroll
            for (int it = 0; it < NRV; ++it) rv[it] = lw[(size_t)min(rb0 * 16 + rowS + 16 * it, K - 1) * ldr + i];
            if (EXP) lv = lse[i];
        }
    };
    auto stage = [&](double* Xb, double* Rb, uint32_t h) {
        if constexpr (EXP == 2) {
            const uint32_t i = h * HT + sR;
            const bool live = i < n;
            double m = -__builtin_inf();
#pragma unroll
            for (int it = 0; it < NRV; ++it) {
                if (cg * NRV + it >= K) rv[it] = -__builtin_inf();          // components beyond K: exp(-inf) = 0
                m = fmax(m, rv[it]);
            }
            m = allreduce_max_bits2345(m, lane);
            double sum = 0.0;
#pragma unroll
            for (int it = 0; it < NRV; ++it) {
                rv[it] = exp_nonpos(rv[it] - m);
                sum += rv[it];
                __builtin_amdgcn_sched_barrier(0);
            }
            sum = allreduce_sum_bits2345(sum, lane);
            const double inv = live ? 1.0 / sum : 0.0;                      // padding samples contribute nothing
#pragma unroll
            for (int it = 0; it < NRV; ++it) Rb[sR * RS + cg * NRV + it] = rv[it] * inv;
            if (cg == 0 && blockIdx.y == 0) {                               // lse = m + log(sum): em_lse_finish_kernel
                lse_out[i] = m;
                ll_out[i] = sum;
            }
        } else {
            const bool live = h * HT + sS < n;
#pragma unroll
            for (int it = 0; it < NRV; ++it) {
                double r = rv[it];
                if (EXP) r = exp_nonpos(r - lv);
                const bool valid = live && (rb0 * 16 + rowS + 16 * it < K);
                Rb[sS * RS + rowS + 16 * it] = valid ? r : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < NXV; ++it) {
            const int j = rowS + 16 * it;
            if (j < d) Xb[sS * XS + j] = xv[it] - shift[j];
        }
        if (wave == 0 && lane < HT) {
            Xb[sS * XS + d] = 1.0;
            Xb[sS * XS + da] = 0.0;
        }
    };

    const uint32_t n_halves = (n + HT - 1) / HT;
    constexpr int tile_doubles = HT * XS + HT * RS;
    int buf = 0;
    if (blockIdx.x < n_halves) prefetch(blockIdx.x);
    for (uint32_t h = blockIdx.x; h < n_halves; h += gridDim.x, buf ^= 1) {
        double* Xb = smem + buf * tile_doubles;
        double* Rb = Xb + HT * XS;
        stage(Xb, Rb, h);
        __syncthreads();
        const uint32_t next = h + gridDim.x;
        prefetch(next < n_halves ? next : h);        // the last iteration re-reads its own half (discarded)
        // ---- contraction: 8 groups of 4 samples. Lane group g = lane >> 4 takes sample sg + 8 g.
        const double* xbase = Xb + 8 * (lane >> 4) * XS;
        const double* rbase = Rb + 8 * (lane >> 4) * RS + (lane & 15);
        __builtin_amdgcn_s_setprio(kMatrixPhasePriority);
#pragma unroll 2
        for (int sg = 0; sg < HT / 4; ++sg) {
            const double* xr = xbase + sg * XS;
            const double* rr = rbase + sg * RS;
            double av[RBW];
#pragma unroll
            for (int r = 0; r < RBW; ++r) av[r] = rr[r * 16];
#pragma unroll
            for (int c = 0; c < CBW - 1; ++c) {
                const double bv = xr[offa[c]] * xr[offb[c]];
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
            }
            constexpr int c = CBW - 1;
            const double bv = xr[offa[c]] * xr[offb[c]];
            if (last_active) {
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    double* out = partials + (size_t)blockIdx.x * KP * FP;
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
            const int cb = (cbg * CBW + c) * NW + wave;
            if (cb < CB_total) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = (rb0 + r) * 16 + (lane >> 4) + 4 * g;
                    out[(size_t)k * FP + cb * 16 + (lane & 15)] = acc[r][c][g];
                }
            }
        }
}

[[maybe_unused]] inline bool half_tiles() { static const bool on = [] { const char* e = std::getenv("MLHIP_MSTATS_HALF"); return e && e[0] == '1'; }(); return on; }

template <int RBW, int CBW>
void launch_half(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
    constexpr int XSD = tile_stride<kMaxDim>();
    const size_t smem = 2 * sizeof(double) * ((size_t)32 * XSD + (size_t)32 * (RBW * 16 + 1));
    const dim3 grid(grid_x, p.n_rbg * p.n_cbg);
#define MLHIP_HALF(E, LO, LL) \
    hipLaunchKernelGGL((em_mstats_half_kernel<RBW, CBW, E>), grid, dim3(512), smem, stream, a.xt, a.ldx, a.n, a.d, padded_dim(a.d), a.shift, \
                       a.lw, a.ldr, a.lse, a.K, p.n_rbg, p.CB, a.partials, p.KP, p.FP, LO, LL)
    if (a.mode == kFromLogRespSelfNorm) MLHIP_HALF(2, a.lse_out, a.ll_out);
    else if (a.mode == kFromLogResp) MLHIP_HALF(1, nullptr, nullptr);
    else MLHIP_HALF(0, nullptr, nullptr);
#undef MLHIP_HALF
}

#endif  // MLHIP_EXPERIMENTS

template <int RBW, int CBW, int DM = kRegDim, int CBT = 0>
void launch_t(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
    constexpr int XSD = tile_stride<DM>();
    const size_t smem = (DM <= kMidDim ? 2 : 1) * sizeof(double) * ((size_t)TS * XSD + (size_t)TS * (RBW * 16 + 1));
    const dim3 grid(grid_x, p.n_rbg * p.n_cbg);
    if (a.mode == kFromLogRespSelfNorm)
        hipLaunchKernelGGL((em_mstats_wide_kernel<RBW, CBW, 2, DM, CBT>), grid, dim3(512), smem, stream, a.xt, a.ldx, a.n, a.d,
                           padded_dim(a.d), a.shift, a.lw, a.ldr, a.lse, a.K, p.n_rbg, p.CB, a.partials, p.KP, p.FP,
                           a.lse_out, a.ll_out);
    else if (a.mode == kFromLogResp)
        hipLaunchKernelGGL((em_mstats_wide_kernel<RBW, CBW, 1, DM, CBT>), grid, dim3(512), smem, stream, a.xt, a.ldx, a.n, a.d,
                           padded_dim(a.d), a.shift, a.lw, a.ldr, a.lse, a.K, p.n_rbg, p.CB, a.partials, p.KP, p.FP,
                           nullptr, nullptr);
    else
        hipLaunchKernelGGL((em_mstats_wide_kernel<RBW, CBW, 0, DM, CBT>), grid, dim3(512), smem, stream, a.xt, a.ldx, a.n, a.d,
                           padded_dim(a.d), a.shift, a.lw, a.ldr, a.lse, a.K, p.n_rbg, p.CB, a.partials, p.KP, p.FP,
                           nullptr, nullptr);
}

}  // namespace

// ---- one code object per (row blocks per workgroup, dimension class): the Makefile compiles this file eight times with
// -DMLHIP_PART=1..8 (parts 1-4: d <= 32 with RBW = part; 5-8: 32 < d <= 128 with RBW = part - 4). A fit uses one K and one d, so it
// touches one or two of them (the sample covariance is a K = 1 call: RBW = 1) and the first use of a shape loads ~0.3 MB of
// kernels instead of the 2.1 MB of all 147 instantiations (round 5: the first call of a shape was dominated by that load).
int MLHIP_PART_FN(launch_wide)(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
    constexpr int R = (MLHIP_PART - 1) % 4 + 1;
    if (p.RBW != R) return -1;
#if MLHIP_PART > 4
#ifdef MLHIP_EXPERIMENTS
#define MLHIP_TRY_HALF(R_, C_) (half_tiles() ? (launch_half<R_, C_>(a, p, grid_x, stream), true) : false)
#else
#define MLHIP_TRY_HALF(R_, C_) false      /* (the half-tile kernel exists in the experiments library only) */
#endif
    // 32 < d <= 128: 38..525 column blocks in column groups of 8 waves x (3, 4 or 5) blocks
#define MLHIP_BIG(C) \
    if (p.CBW == C) { \
        if (a.d <= kMidDim) launch_t<R, C, kMidDim>(a, p, grid_x, stream); else if (!MLHIP_TRY_HALF(R, C)) launch_t<R, C, kMaxDim>(a, p, grid_x, stream); \
    } else
    MLHIP_BIG(3) MLHIP_BIG(4) MLHIP_BIG(5)
    { return -1; }
#undef MLHIP_BIG
    return grid_x;
#else
    // balanced dealing of (column block, row block) units where whole column blocks leave the waves unevenly loaded
    // (MLHIP_MSTATS_BALANCED=0: off)
    static const bool balanced = [] { const char* e = std::getenv("MLHIP_MSTATS_BALANCED"); return !(e && e[0] == '0'); }();
    if (balanced && p.n_rbg == 1 && p.n_cbg == 1) {
        if constexpr (R == 4) {
            if (p.CB == 6) { launch_t<4, 1, kRegDim, 6>(a, p, grid_x, stream); return grid_x; }      // d = 12
            if (p.CB == 10) { launch_t<4, 2, kRegDim, 10>(a, p, grid_x, stream); return grid_x; }    // d = 16
        }
        if constexpr (R == 3) {
            if (p.CB == 10) { launch_t<3, 2, kRegDim, 10>(a, p, grid_x, stream); return grid_x; }
            if (p.CB == 6) { launch_t<3, 1, kRegDim, 6>(a, p, grid_x, stream); return grid_x; }
        }
        // (K <= 32 at d = 16: the plain form with two workgroups per CU -- the balanced one needs 130 registers -- unless the plan
        //  kept one workgroup per CU)
        if constexpr (R == 2) {
            if (p.CB == 10 && p.wg_per_cu == 1) { launch_t<2, 2, kRegDim, 10>(a, p, grid_x, stream); return grid_x; }
        }
    }
#define MLHIP_CASE(C) \
    if (p.CBW == C) { launch_t<R, C>(a, p, grid_x, stream); } else
    MLHIP_CASE(1) MLHIP_CASE(2) MLHIP_CASE(3) MLHIP_CASE(4) MLHIP_CASE(5)
    { return -1; }
#undef MLHIP_CASE
    return grid_x;
#endif
}

#if MLHIP_PART == 1
int launch_wide_part2(const MstatsArgs&, const Plan&, int, hipStream_t);
int launch_wide_part3(const MstatsArgs&, const Plan&, int, hipStream_t);
int launch_wide_part4(const MstatsArgs&, const Plan&, int, hipStream_t);
int launch_wide_part5(const MstatsArgs&, const Plan&, int, hipStream_t);
int launch_wide_part6(const MstatsArgs&, const Plan&, int, hipStream_t);
int launch_wide_part7(const MstatsArgs&, const Plan&, int, hipStream_t);
int launch_wide_part8(const MstatsArgs&, const Plan&, int, hipStream_t);

int launch_wide(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
    const bool big = a.d > kRegDim;
    switch (p.RBW) {
    case 1: return big ? launch_wide_part5(a, p, grid_x, stream) : launch_wide_part1(a, p, grid_x, stream);
    case 2: return big ? launch_wide_part6(a, p, grid_x, stream) : launch_wide_part2(a, p, grid_x, stream);
    case 3: return big ? launch_wide_part7(a, p, grid_x, stream) : launch_wide_part3(a, p, grid_x, stream);
    case 4: return big ? launch_wide_part8(a, p, grid_x, stream) : launch_wide_part4(a, p, grid_x, stream);
    default: return -1;
    }
}
#endif

}  // namespace mstats
}  // namespace mlhip
