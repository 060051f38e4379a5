// M-step sufficient statistics for SMALL dimensions (d <= 9: at most 4 column blocks of 16 features) -- see em_mstats.hip
// for the formulation (one fp64-MFMA GEMM stats[K x F] = R^T Phi, Phi generated in registers).
//
// With so few features the GEMM is tiny next to the bytes it consumes (X, the K log-responsibility columns, LSE): the
// kernel has to run at HBM speed, and a workgroup-wide tile with a barrier per 64 samples cannot (1.3 us per tile and CU
// at d = 2..4, 2-3x off the HBM bound). Here every WAVE is on its own: it owns a stream of 64-sample tiles, stages them in
// a private LDS region (no workgroup barrier in the loop), holds ALL column blocks and up to 4 row blocks of accumulators,
// and prefetches the next block of responsibility rows / the next tile while the matrix cores work on the current one.
// The exp(lw - lse) normalisation is applied once per (sample, component) while staging, and skipped for padding rows.
// The four waves of a workgroup fold their accumulators into the workgroup's partial block one after the other (fixed
// order: the result is bit-reproducible).
#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"

#ifndef SMALL_STATS_UNROLL
#define SMALL_STATS_UNROLL 16   // the 16 sample groups of a tile, all of them [r5] (d = 8, K = 32: 1.73 -> 1.69 ms)
#endif

namespace mlhip {
namespace mstats {
namespace {

constexpr int XSS = 11;   // LDS row stride of the sample tile: d + 1 coordinates + zero slot <= 11 for d <= 9, odd
constexpr int RSS = 17;   // LDS row stride of one 16-component responsibility block, odd

template <int RBW, int CB, bool EXP>
__global__ __launch_bounds__(256, 2) void em_mstats_small_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ shift,
    const double* __restrict__ lw, size_t ldr, const double* __restrict__ lse, int K, int F,
    double* __restrict__ partials, int KP, int FP)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* Xw = smem + (size_t)wave * (TS * XSS + TS * RSS);   // this wave's private tiles
    double* Rw = Xw + TS * XSS;
    const int da = d + 1;
    const int rb0 = blockIdx.y * RBW;                           // first 16-component row block of this workgroup

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
    // operand bases: lane group g = lane >> 4 takes sample sg + 16 g of the tile (rows 16 apart sit 32 banks apart)
    const double* xbase = Xw + 16 * (lane >> 4) * XSS;
    const double* rbase = Rw + 16 * (lane >> 4) * RSS + (lane & 15);

    double rv[16], xv[9], lv = 0.0;
    auto load_rows = [&](uint32_t tile, int rb) {               // 16 responsibility rows of block rb for sample `lane`
        const uint32_t i = tile * TS + lane;                    // < n_pad: inside the allocation
#pragma unroll
        for (int it = 0; it < 16; ++it) rv[it] = lw[(size_t)min((rb0 + rb) * 16 + it, K - 1) * ldr + i];
    };
    auto load_sample = [&](uint32_t tile) {
        const uint32_t i = tile * TS + lane;
#pragma unroll
        for (int j = 0; j < 9; ++j) xv[j] = xt[(size_t)min(j, d - 1) * ldx + i];
        if (EXP) lv = lse[i];
    };

    uint32_t tile = blockIdx.x * 4 + wave;
    if (tile < n_tiles) { load_sample(tile); load_rows(tile, 0); }
    for (; tile < n_tiles; tile += stride) {
        const bool live = tile * TS + lane < n;
        const uint32_t next = tile + stride < n_tiles ? tile + stride : tile;
        // ---- sample tile -> LDS (the previous tile's readers are this same wave: program order suffices)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j < d) Xw[lane * XSS + j] = xv[j] - shift[j];
        Xw[lane * XSS + d] = 1.0;
        Xw[lane * XSS + da] = 0.0;
        const double lcur = lv;
        load_sample(next);                                      // in flight during the whole tile
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            // ---- responsibility block rb -> LDS (exp once per (sample, component); padding rows are zero)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it4 = 0; it4 < 16; it4 += 4) {
                // guards per group of 4 rows (wave-uniform): rows beyond K inside a live group were loaded from row K-1
                // (clamped) and are zeroed by the select
                if ((rb0 + rb) * 16 + it4 < K) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        double r = rv[it4 + u];
                        if (EXP) r = exp_nonpos(r - lcur);
                        Rw[lane * RSS + it4 + u] = (live && (rb0 + rb) * 16 + it4 + u < K) ? r : 0.0;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) Rw[lane * RSS + it4 + u] = 0.0;
                }
            }
            // next block of rows (or the first block of the next tile): in flight during the MFMA phase
            if (rb + 1 < RBW) load_rows(tile, rb + 1); else load_rows(next, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
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
        }
    }

    // ---- epilogue: the waves fold their accumulators into partials[blockIdx.x] one after the other (fixed order).
    // C/D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* out = partials + (size_t)blockIdx.x * KP * FP;
    for (int w = 0; w < 4; ++w) {
        if (w == wave) {
#pragma unroll
            for (int r = 0; r < RBW; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int k = (rb0 + r) * 16 + (lane >> 4) + 4 * g;
                        double* p = out + (size_t)k * FP + c * 16 + (lane & 15);
                        *p = (w == 0 ? 0.0 : *p) + acc[r][c][g];
                    }
        }
        __syncthreads();
    }
}

template <int RBW, int CB>
void launch_t(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
    const size_t smem = 4 * sizeof(double) * ((size_t)TS * XSS + (size_t)TS * RSS);
    const dim3 grid(grid_x, p.n_rbg);
    const int F = stats_count(a.d);
    if (a.mode == kFromLogResp)
        hipLaunchKernelGGL((em_mstats_small_kernel<RBW, CB, true>), grid, dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                           a.lw, a.ldr, a.lse, a.K, F, a.partials, p.KP, p.FP);
    else
        hipLaunchKernelGGL((em_mstats_small_kernel<RBW, CB, false>), grid, dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d, a.shift,
                           a.lw, a.ldr, a.lse, a.K, F, a.partials, p.KP, p.FP);
}

}  // namespace

int launch_small(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
#define MLHIP_CASE(R, C) \
    if (p.RBW == R && p.CB == C) { launch_t<R, C>(a, p, grid_x, stream); } else
    MLHIP_CASE(1, 1) MLHIP_CASE(1, 2) MLHIP_CASE(1, 3) MLHIP_CASE(1, 4)
    MLHIP_CASE(2, 1) MLHIP_CASE(2, 2) MLHIP_CASE(2, 3) MLHIP_CASE(2, 4)
    MLHIP_CASE(4, 1) MLHIP_CASE(4, 2) MLHIP_CASE(4, 3) MLHIP_CASE(4, 4)
    { return -1; }
#undef MLHIP_CASE
    return grid_x;
}

}  // namespace mstats
}  // namespace mlhip
