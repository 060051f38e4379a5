// E-step of Gaussian-mixture EM on the gfx950 fp64 matrix cores with ELEMENT-BLOCK triangular work (dimensions 12..128)
// -- replaces EM::expectation_step (reference ML/EM.cpp:190-219) and its xAx_symmetric calls (ML/LinearAlgebra.cpp:8-31).
//
// Same arithmetic as the 16x16x4 variant it superseded (experiments/em_estep_mfma16.hip) (z = x - mu_k, y = W_k z with W_k = L_k^-1 lower triangular, q = |y|^2,
// lw = log pi_k - sum log L_jj - q/2, online log-sum-exp), but on v_mfma_f64_4x4x4_4b_f64: one instruction multiplies
// FOUR independent 4x4 blocks, D_b[4 rows][4 samples] += A_b[4 x 4] * B_b[4 x 4 samples]. With the same 4x4 block of W in
// all four A blocks and four different sample quads in B, one instruction advances 4 rows of y for 16 samples, and only
// the 4x4 blocks of W on or below the diagonal are ever issued: Q(Q+1)/2 = 36 of 64 at d = 32, i.e. 1152 flop per
// (sample, component) instead of the 1536 of 16x16 blocks -- the matrix-core rate is the same for both shapes
// (tools/microbench_fp64: 75.9 vs 72.1 TFLOP/s).
//
// Lane layout of the instruction (probed on the hardware, tools/probe_mfma4.hip):
//   A[b][i][k] <- lane 16k + 4b + i,   B[b][k][j] <- lane 16k + 4b + j,   D[b][i][j] -> lane 16i + 4b + j.
// With j + 4b = sample-in-16 this is the B / D layout of the 16x16x4 kernel (lane & 15 = sample, lane >> 4 = k or row),
// so the coordinates stay in VGPRs as xb[column quad][sample block], every accumulator is ONE double per lane
// (acc[sample block][row quad]) and the |y|^2 reduce-scatter / log-domain epilogue are unchanged.
//
// Two compile-time options take VALU work off the fp64 pipe the MFMAs share (round 1: ~118 VALU operations per 64 samples
// and component next to 144 MFMAs):
//   FOLD  y = W (x - s) - W (mu - s) with the second term as the accumulator initialiser (s = the data's shift, x - s formed
//         once per sample group): no per-component subtraction of the mean (32 operations). It costs log2(|W (mu - s)|) bits
//         of y, so the host only selects it while every |W_k (mu_k - s)| entry is below a limit (layout.hpp
//         kEstepFoldLimit); beyond that the exact form (x - mu_k first) runs.
//   !LSE  no online log-sum-exp, only lw is written: the statistics kernel normalises a sample's K log-responsibilities
//         itself while staging its tile (em_mstats_wide.hip, one exp per (sample, component) instead of one in each
//         kernel) and produces lse and the log-likelihood sum. Used whenever K fits one row-block group of that kernel.
#include <cstdlib>

#include "device.hpp"
#include "exp_nonpos.hpp"

namespace mlhip {
namespace {

template <int D> struct Blocks {
    static constexpr int Q = D / 4;                   // 4-wide quads of rows / columns
    static constexpr int NB = Q * (Q + 1) / 2;        // blocks on or below the diagonal
    static constexpr int PS = NB * 16 + 2 * D + 1;    // doubles per component record (layout.hpp estep_mfma4_param_stride)
    // step t (column-quad major): t -> (C, R), R = C..Q-1
    static constexpr int start_of(int C) { return C * Q - C * (C - 1) / 2; }
    static constexpr int C(int t) { int c = 0; while (c + 1 < Q && start_of(c + 1) <= t) ++c; return c; }
    static constexpr int R(int t) { return C(t) + (t - start_of(C(t))); }
    static constexpr bool first_of_C(int t) { return t == start_of(C(t)); }
    // rolling prefetch window: the largest divisor of NB that is <= 9
    static constexpr int window() { for (int w = 9; w > 1; --w) if (NB % w == 0) return w; return 1; }
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/// (As in experiments/em_estep_mfma16.hip:) group g = lane>>4 ends with the sum over groups of v[g] (v_permlane16/32_swap reduce-scatter).
__device__ __forceinline__ double reduce_scatter_groups(double v0, double v1, double v2, double v3)
{
    auto swap16 = [](double& a, double& b) {
        const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    };
    auto swap32 = [](double& a, double& b) {
        const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    };
    swap16(v0, v1);
    swap16(v2, v3);
    double t01 = v0 + v1, t23 = v2 + v3;
    swap32(t01, t23);
    return t01 + t23;
}

/// SB = 16-sample blocks per wave (4: 64 samples per wave, 4 waves per workgroup, 2 waves per SIMD -- d <= 32; 2: 32
/// samples per wave, 8 waves per workgroup, half the coordinate/accumulator registers -- 32 < d <= 64, or
/// MLHIP_ESTEP_SB=2; 1: 16 samples per wave, 8 waves of 512 threads -- 64 < d <= 128). A workgroup covers 256 samples
/// per component sweep for SB = 4 and 2 (128 for SB = 1); the component record is staged once per sweep.
template <int SB> constexpr int default_waves() { return SB == 1 ? 8 : 16 / SB; }

template <int D, int SB, bool FOLD, bool LSE, int NW = default_waves<SB>()>
__global__ __launch_bounds__(64 * NW, D <= 32 ? 8 / SB : 2) void em_estep_mfma4_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, uint32_t n_groups, const double* __restrict__ params, int K,
    const double* __restrict__ shift, double* __restrict__ lw_out, size_t ldr, double* __restrict__ lse_out,
    double* __restrict__ ll_partials)
{
    using B = Blocks<D>;
    constexpr int Q = B::Q, NB = B::NB, PS = B::PS;
    constexpr int NT = 64 * NW, NWV = NW;          // threads / waves per workgroup
    constexpr int GS = 16 * SB;                    // samples per wave
    constexpr int W = 4;                           // LDS read-ahead window (blocks)
    constexpr int NLD = (PS + NT - 1) / NT;        // doubles of a record each thread moves to LDS
    __shared__ double red[NWV];
    extern __shared__ __attribute__((aligned(16))) double recs_dyn[];   // [2][NLD * NT]: the component record, staged
    double (*recs)[NLD * NT] = reinterpret_cast<double (*)[NLD * NT]>(recs_dyn);   // once per workgroup, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, s = lane & 15;
    const int aoff = g * 4 + (lane & 3);          // A operand: entry [k = lane>>4][i = lane&3] of a 16-double block
    const bool owner = g < SB;                    // lane (g, s) owns sample 16g + s of the wave's group
    double ll_acc = 0.0;

    // The waves of a workgroup walk the components in lockstep (they share the staged record), each on its own
    // GS-sample group: a workgroup iteration covers NWV consecutive groups = 256 samples (n_pad is a multiple of 256).
    auto process = [&](uint32_t grp, int k_begin, int k_end) {
        const uint32_t base = grp * GS;
        // coordinates in B-operand layout: xb[C][sb] = x[dim 4C + g][sample base + 16sb + s]
        double xb[Q][SB];
        {
            // (The row pointer is made opaque per tile: left to itself the compiler hoists all Q x SB 64-bit load addresses out of the
            // tile loop and carries them in registers -- 64 to 128 of them at d > 64, spilled to scratch around every tile in round 4.)
            size_t first = (size_t)g * ldx + base + s;      // (the OFFSET is made opaque, not the pointer: loads through a laundered pointer
            asm volatile("" : "+v"(first));                  //  become flat loads, which also count on the LDS counter)
            const double* row = xt + first;
#pragma unroll
            for (int C = 0; C < Q; ++C)
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) xb[C][sb] = row[(size_t)(4 * C) * ldx + 16 * sb];
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int C = 0; C < Q; ++C) {
                const double sh = shift[4 * C + g];       // zero-padded to D entries
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) xb[C][sb] -= sh;
            }
        }

        double m = -__builtin_inf(), ssum = 0.0;

        // first record -> its LDS buffer (the barrier at the top of the component loop publishes it)
        double stage[NLD];
#pragma unroll
        for (int it = 0; it < NLD; ++it) stage[it] = params[(size_t)k_begin * PS + min(tid + NT * it, PS - 1)];
        __syncthreads();                            // everyone is done with the previous group's last record
#pragma unroll
        for (int it = 0; it < NLD; ++it) recs[k_begin & 1][tid + NT * it] = stage[it];

        for (int k = k_begin; k < k_end; ++k) {
            const double* __restrict__ rec = recs[k & 1];
            __syncthreads();                        // record k visible; every wave has finished component k-1
            // record k+1: global -> registers now (in flight during the MFMA phase), registers -> LDS at the end
            const double* __restrict__ nxt = params + (size_t)(k + 1 < k_end ? k + 1 : k) * PS;
#pragma unroll
            for (int it = 0; it < NLD; ++it) stage[it] = nxt[min(tid + NT * it, PS - 1)];

            const double coef = rec[NB * 16 + 2 * D];
            double aw[W], acc[SB][Q], z[SB];
#pragma unroll
            for (int t = 0; t < W && t < NB; ++t) aw[t] = rec[t * 16 + aoff];
            // Nested static loops (column quad C, then row quad R >= C): every index below is a compile-time constant
            // after unrolling, so the accumulators and the window stay in registers.
            // The wave in its matrix phase gets the issue priority over the SIMD's other wave, which is in its VALU phase
            // (squares, reduce-scatter, epilogue) or waiting at the barrier: the MFMA stream is not broken up by the
            // neighbour's VALU instructions, which fit into its shadow instead. Measured: d = 32, K = 64 12.35 -> 12.0 ms;
            // -1.5 % at d = 24 / 28, neutral at d = 20 and from d = 48 on, but +3 .. 4 % at d = 12 / 16, where the matrix phase of a
            // component is 6 - 10 blocks short and the neighbour's epilogue is most of the work: not used there.
            constexpr bool kPrioritise = D >= 20;
            if constexpr (kPrioritise) __builtin_amdgcn_s_setprio(kMatrixPhasePriority);
#pragma unroll
            for (int C = 0; C < Q; ++C) {
                // the record's two vectors: the mean (exact form) / -W (mu - shift) (FOLD: the accumulator initialiser)
                const double mu = FOLD ? 0.0 : rec[NB * 16 + 4 * C + g];
#pragma unroll
                for (int R = C; R < Q; ++R) {
                    const int t = C * Q - C * (C - 1) / 2 + (R - C);     // step index in column-quad-major order
                    const double a = aw[t % W];
                    if (R == C) {
#pragma unroll
                        for (int sb = 0; sb < SB; ++sb) z[sb] = FOLD ? xb[C][sb] : xb[C][sb] - mu;
                    }
                    if (t + W < NB) aw[t % W] = rec[(t + W) * 16 + aoff];   // refill the slot (LDS broadcast read)
                    double init = 0.0;
                    if constexpr (FOLD) { if (C == 0) init = rec[NB * 16 + D + 4 * R + g]; }
#pragma unroll
                    for (int sb = 0; sb < SB; ++sb)
                        acc[sb][R] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z[sb], C == 0 ? init : acc[sb][R], 0, 0, 0);
                    // Pin the block order (column-quad major, the sample blocks back to back); the same pinning as in experiments/em_estep_mfma16.hip.
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (kPrioritise) __builtin_amdgcn_s_setprio(0);
            double qs[SB];
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                double t2 = 0.0;
#pragma unroll
                for (int R = 0; R < Q; ++R) t2 = __builtin_fma(acc[sb][R], acc[sb][R], t2);
                qs[sb] = t2;
            }
            double q;
            if constexpr (SB == 1) {
                // one block: the four lane groups hold the four row residues of the same 16 samples
                const double t = qs[0] + __shfl_xor(qs[0], 16, 64);
                q = t + __shfl_xor(t, 32, 64);
            } else if constexpr (SB == 4) {
                q = reduce_scatter_groups(qs[0], qs[1], qs[2], qs[3]);   // lane (g, s) gets sample 16g + s
            } else {
                // two blocks: rows g, g^1 exchange so that even rows hold block 0 and odd rows block 1, then the two
                // halves are summed; lanes (g, s) with g < 2 own sample 16g + s, g >= 2 hold duplicates.
                double v0 = qs[0], v1 = qs[1];
                const unsigned alo = __double2loint(v0), ahi = __double2hiint(v0), blo = __double2loint(v1), bhi = __double2hiint(v1);
                const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
                v0 = __hiloint2double((int)hi[0], (int)lo[0]);
                v1 = __hiloint2double((int)hi[1], (int)lo[1]);
                const double t = v0 + v1;                  // even rows: block 0 over the pair, odd rows: block 1
                q = t + __shfl_xor(t, 32, 64);
            }
            const double lw = __builtin_fma(-0.5, q, coef);
            if (owner) lw_out[(size_t)k * ldr + base + lane] = lw;
            if constexpr (LSE) {
                // exp(t) is exactly 0 in fp64 for t < -745.2: when that holds for the whole wave the update would add 0 to
                // every ssum and leave every m unchanged, so it is skipped (bit-identical result, one exp saved).
                if (!__all(lw - m < -746.0)) {
                    // (mixing weight 0: lw = -inf adds exp(-inf) = 0 as the reference's `column *= 0` does, ML/EM.cpp:209; -inf - -inf
                    // would poison the sum while the running maximum is still -inf; a genuinely NaN lw stays a NaN)
                    const double e = exp_nonpos(lw == -HUGE_VAL ? -HUGE_VAL : -fabs(lw - m));
                    const bool up = lw > m;
                    ssum = up ? __builtin_fma(ssum, e, 1.0) : ssum + e;
                    m = up ? lw : m;
                }
            }
            // publish record k+1 in the other buffer: nobody reads it now (last read during k-1, before this
            // iteration's barrier), the next iteration's barrier makes it visible.
#pragma unroll
            for (int it = 0; it < NLD; ++it) recs[(k + 1) & 1][tid + NT * it] = stage[it];
        }
        if constexpr (LSE) {
            const double lse = m + log(ssum);
            if (owner) {
                lse_out[base + lane] = lse;
                if (base + lane < n) ll_acc += lse;
            }
        }
    };
    // Whole tiles (NWV consecutive groups = 256 samples, all K components) round after round; what is left for the last, partial
    // round is cut into (tile, component range) units when the log-sum-exp is formed elsewhere (!LSE: the components of a sample
    // are then independent here): with R < gridDim tiles left, R kTailChunks units of K / kTailChunks components fill the
    // workgroups evenly instead of leaving gridDim - R of them idle for a whole tile (one 8-GPU shard of the headline: 4 883
    // tiles over 512 workgroups = 9.54 rounds, 10 without the split, 9.63 with it).
    constexpr int kTailChunks = 8;
    const uint32_t n_tiles = n_groups / NWV;                       // (n_pad is a multiple of the workgroup's sweep)
    const bool split_tail = !LSE && K >= 2 * kTailChunks;
    const uint32_t full_tiles = split_tail ? n_tiles / gridDim.x * gridDim.x : n_tiles;
    for (uint32_t tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) process(tile * NWV + wave, 0, K);
    if (split_tail) {
        const uint32_t units = (n_tiles - full_tiles) * kTailChunks;
        for (uint32_t u = blockIdx.x; u < units; u += gridDim.x) {
            const int c = (int)(u % kTailChunks);
            process((full_tiles + u / kTailChunks) * NWV + wave, c * K / kTailChunks, (c + 1) * K / kTailChunks);
        }
    }
    if constexpr (LSE) {
        ll_acc = wave_sum(ll_acc);
        if (lane == 0) red[wave] = ll_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t += red[w];
            ll_partials[blockIdx.x] = t;
        }
    }
}

template <int D, int SB, bool FOLD, bool LSE, int NW = default_waves<SB>()>
int launch_sb(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    // the waves of a workgroup share barriers inside the component loop: a workgroup sweep must cover a whole number of
    // 256-sample tiles, or the last sweep would leave some waves outside the loop
    static_assert((16 * SB * NW) % kSampleTile == 0 || kSampleTile % (16 * SB * NW) == 0, "sweep vs tile");
    static_assert(16 * SB * NW <= kSampleTile, "a workgroup sweep may not exceed the padding granule of N");
    constexpr int NT = 64 * NW, NWV = NW, GS = 16 * SB;
    constexpr int NLD = (Blocks<D>::PS + NT - 1) / NT;
    const size_t smem = sizeof(double) * 2 * NLD * NT;
    const uint32_t n_pad = padded_samples(a.n);
    const uint32_t n_groups = n_pad / GS;
    uint32_t grid = (n_groups + NWV - 1) / NWV;
    uint32_t per_cu = 2 * default_waves<SB>() / NW;                          // the CU's resident workgroups, persistent
    if constexpr (SB == 4 && D <= 20) {
        // small d: a component is 6 - 21 blocks short, the per-component barrier and the epilogue weigh more, and the kernel needs
        // few registers (92 - 138): as many workgroups per CU as fit, up to 4 [r3] (MLHIP_ESTEP_WGS=2: the two of larger d; at d = 24
        // three fit, the E-step gains 1 % and the statistics kernel behind it loses 2 %: not used there)
        static const int fit = [] {
            int blocks = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, em_estep_mfma4_kernel<D, SB, FOLD, LSE, NW>, NT, sizeof(double) * 2 * NLD * NT) != hipSuccess) blocks = 2;
            const char* e = ab_env("MLHIP_ESTEP_WGS");
            const int want = e ? std::atoi(e) : 4;
            return blocks < 2 ? 2 : (blocks > want ? want : blocks);
        }();
        if ((uint32_t)fit > per_cu) per_cu = (uint32_t)fit;
    }
    const uint32_t cap = (uint32_t)num_cus * per_cu;
    if (grid > cap) grid = cap;
    if (grid > (uint32_t)a.n_ll_partials) grid = (uint32_t)a.n_ll_partials;
    hipLaunchKernelGGL((em_estep_mfma4_kernel<D, SB, FOLD, LSE, NW>), dim3(grid), dim3(NT), smem, stream, a.xt, a.ldx, a.n, n_groups,
                       a.params, a.K, a.shift, a.lw, a.ldr, a.lse, a.ll_partials);
    return (int)grid;
}

template <int D, int SB>
int launch_v(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    if constexpr (D <= kRegDim) {
        if (a.fold) return a.with_lse ? launch_sb<D, SB, true, true>(a, num_cus, stream) : launch_sb<D, SB, true, false>(a, num_cus, stream);
    }
    return a.with_lse ? launch_sb<D, SB, false, true>(a, num_cus, stream) : launch_sb<D, SB, false, false>(a, num_cus, stream);
}

template <int D>
int launch_t(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    // 64 < d: one sample block per wave (16 samples): 2D doubles per lane; 32 < d <= 64: two (D coordinate + D accumulator
    // doubles per lane pair do not fit otherwise); d <= 32: four (64 samples per wave)
    // (Round 5 measured TWO blocks per wave at one wave per SIMD for 64 < d <= 128 -- 512 registers, half the LDS reads of W per matrix
    // instruction: no faster, E-step 10.77 against 10.58 ms at N = 1M, d = 128, K = 32, 6.09 / 5.95 at d = 96, 3.70 / 3.60 at d = 72; not kept.)
    if constexpr (D > 64) return launch_v<D, 1>(a, num_cus, stream);
    else if constexpr (D > 32) return launch_v<D, 2>(a, num_cus, stream);
    else return launch_v<D, 4>(a, num_cus, stream);
}

static_assert(Blocks<32>::NB == 36 && Blocks<12>::NB == 6, "block count");
static_assert(Blocks<32>::C(0) == 0 && Blocks<32>::R(7) == 7 && Blocks<32>::C(8) == 1 && Blocks<32>::R(8) == 1 && Blocks<32>::C(35) == 7, "block order");

}  // namespace

int launch_em_estep_mfma4(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    switch (a.D) {
    case 12: return launch_t<12>(a, num_cus, stream);
    case 16: return launch_t<16>(a, num_cus, stream);
    case 20: return launch_t<20>(a, num_cus, stream);
    case 24: return launch_t<24>(a, num_cus, stream);
    case 28: return launch_t<28>(a, num_cus, stream);
    case 32: return launch_t<32>(a, num_cus, stream);
    case 40: return launch_t<40>(a, num_cus, stream);
    case 48: return launch_t<48>(a, num_cus, stream);
    case 56: return launch_t<56>(a, num_cus, stream);
    case 64: return launch_t<64>(a, num_cus, stream);
    case 72: return launch_t<72>(a, num_cus, stream);
    case 80: return launch_t<80>(a, num_cus, stream);
    case 88: return launch_t<88>(a, num_cus, stream);
    case 96: return launch_t<96>(a, num_cus, stream);
    case 104: return launch_t<104>(a, num_cus, stream);
    case 112: return launch_t<112>(a, num_cus, stream);
    case 120: return launch_t<120>(a, num_cus, stream);
    case 128: return launch_t<128>(a, num_cus, stream);
    default: return -1;
    }
}

}  // namespace mlhip
