// M-step sufficient statistics, workgroup-tile variant for few column blocks (4 waves of a 256-thread workgroup split the
// column blocks). EXPERIMENT ONLY: superseded by em_mstats_small.hip (d <= 9) and em_mstats_wide.hip (d >= 8); built with
// `make EXPERIMENTS=1` and selected with MLHIP_MSTATS=n for A/B runs. Formulation: see em_mstats.hip.
#include "../em_mstats_common.hpp"

namespace mlhip {
namespace {

using namespace mstats;


/// EXP = true : r = exp(lw - lse)   (log-responsibilities left by an E-step)
/// EXP = false: r = lw               (plain responsibilities: caller-given, one-hot from labels, or all ones)
template <int RBW, int CBW, bool EXP>
__global__ __launch_bounds__(256, 2) void em_mstats_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, int D, const double* __restrict__ shift,
    const double* __restrict__ lw, size_t ldr, const double* __restrict__ lse, int K, int n_rbg, int CB_total,
    double* __restrict__ partials, int KP, int FP)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int da = d + 1;          // augmented length; slot `da` of every row holds 0 for padding columns
    constexpr int RS = RBW * 16 + 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rbg = blockIdx.y % n_rbg, cbg = blockIdx.y / n_rbg;
    const int rb0 = rbg * RBW;                       // first 16-component row block of this workgroup
    const int cb0 = (cbg * 4 + wave) * CBW;          // first 16-feature column block of this wave
    const int F = da * (da + 1) / 2;

    int offa[CBW], offb[CBW];
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
        int a, b;
        feature_pair((cb0 + c) * 16 + (lane & 15), (cb0 + c) < CB_total ? F : 0, da, a, b);
        offa[c] = a;
        offb[c] = b;
    }

    d4 acc[RBW][CBW];
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[r][c] = d4{0.0, 0.0, 0.0, 0.0};

    // ---- software pipeline over tiles: the global loads of tile t+1 are in flight during the MFMA phase of tile t;
    // the LDS tiles are double-buffered so one barrier per tile suffices (a wave can only start overwriting buffer b
    // after the barrier of the tile in between, which every wave reaches after it finished reading buffer b).
    // Staging role of a thread: sample sS of the tile, element class qS = its wave index (256 = 4 * TS). All loads are
    // unconditional (row indices clamped into the allocation, the value is discarded when staged) so that they are
    // issued back to back and nothing waits for them before the MFMA loop.
    const int sS = tid & (TS - 1), qS = __builtin_amdgcn_readfirstlane(tid / TS);
    double xv[kRegDim / 4], rv[RBW * 4], lv = 0.0;

    auto prefetch = [&](uint32_t tile) {
        const uint32_t i = tile * TS + sS;           // < n_pad: always inside the allocation
#pragma unroll
        for (int it = 0; it < kRegDim / 4; ++it) xv[it] = xt[(size_t)min(qS + 4 * it, D - 1) * ldx + i];
#pragma unroll
        for (int it = 0; it < RBW * 4; ++it) rv[it] = lw[(size_t)min(rb0 * 16 + qS + 4 * it, K - 1) * ldr + i];
        if (EXP) lv = lse[i];
    };
    auto stage = [&](double* Xb, double* Rb, uint32_t tile) {
        const bool live = tile * TS + sS < n;
#pragma unroll
        for (int it = 0; it < RBW * 4; ++it) {
            double r = rv[it];
            if (EXP) {
                r = exp(r - lv);
                __builtin_amdgcn_sched_barrier(0);     // one exp at a time: keeps its temporaries from piling up
            }
            const bool valid = live && (rb0 * 16 + qS + 4 * it < K);
            Rb[sS * RS + qS + 4 * it] = valid ? r : 0.0;
        }
#pragma unroll
        for (int it = 0; it < kRegDim / 4; ++it) {
            const int j = qS + 4 * it;
            if (j < d) Xb[sS * XS + j] = xv[it] - shift[j];   // wave-uniform index: scalar load
        }
        if (qS == 0) {
            Xb[sS * XS + d] = 1.0;
            Xb[sS * XS + da] = 0.0;
        }
    };

    const uint32_t n_tiles = (n + TS - 1) / TS;
    const int tile_doubles = TS * XS + TS * RS;
    int buf = 0;
    if (blockIdx.x < n_tiles) prefetch(blockIdx.x);
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        double* Xb = smem + buf * tile_doubles;
        double* Rb = Xb + TS * XS;
        stage(Xb, Rb, tile);
        __syncthreads();
        const uint32_t next = tile + gridDim.x;
        prefetch(next < n_tiles ? next : tile);      // the last iteration re-reads its own tile (discarded)
        // ---- contraction: 16 groups of 4 samples. Lane group g = lane>>4 takes sample sg + 16 g: rows 16 apart are
        // 32 banks apart for both tiles (odd strides 35 / 33 doubles), so the two rows of a half-wave never collide.
        // Not unrolled: the accumulators leave too few registers for a second set of operands (unrolling spills).
        const double* xbase = Xb + 16 * (lane >> 4) * XS;
        const double* rbase = Rb + 16 * (lane >> 4) * RS + (lane & 15);
#pragma unroll 1
        for (int sg = 0; sg < TS / 4; ++sg) {
            const double* xr = xbase + sg * XS;
            const double* rr = rbase + sg * RS;
            double av[RBW];
#pragma unroll
            for (int r = 0; r < RBW; ++r) av[r] = rr[r * 16];
#pragma unroll
            for (int c = 0; c < CBW; ++c) {
                const double bv = xr[offa[c]] * xr[offb[c]];
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r], bv, acc[r][c], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: partials[blockIdx.x][k][f]; C/D layout of v_mfma_f64_16x16x4: col = lane&15, row = (lane>>4) + 4*reg
    double* out = partials + (size_t)blockIdx.x * KP * FP;
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
            const int cb = cb0 + c;
            if (cb < CB_total && (rb0 + r) * 16 < KP) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = (rb0 + r) * 16 + (lane >> 4) + 4 * g;
                    out[(size_t)k * FP + cb * 16 + (lane & 15)] = acc[r][c][g];
                }
            }
        }
}

template <int RBW, int CBW>
void launch_t(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
    const size_t smem = 2 * sizeof(double) * ((size_t)TS * XS + (size_t)TS * (RBW * 16 + 1));   // double-buffered
    const dim3 grid(grid_x, p.n_rbg * p.n_cbg);
    if (a.mode == kFromLogResp)
        hipLaunchKernelGGL((em_mstats_kernel<RBW, CBW, true>), grid, dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d,
                           padded_dim(a.d), a.shift, a.lw, a.ldr, a.lse, a.K, p.n_rbg, p.CB, a.partials, p.KP, p.FP);
    else
        hipLaunchKernelGGL((em_mstats_kernel<RBW, CBW, false>), grid, dim3(256), smem, stream, a.xt, a.ldx, a.n, a.d,
                           padded_dim(a.d), a.shift, a.lw, a.ldr, a.lse, a.K, p.n_rbg, p.CB, a.partials, p.KP, p.FP);
}

}  // namespace

namespace mstats {

int launch_narrow(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream)
{
#define MLHIP_CASE(R, C) \
    if (p.RBW == R && p.CBW == C) { launch_t<R, C>(a, p, grid_x, stream); } else
    MLHIP_CASE(1, 1) MLHIP_CASE(1, 2) MLHIP_CASE(2, 1) MLHIP_CASE(2, 2) MLHIP_CASE(4, 1) MLHIP_CASE(4, 2)
    { return -1; }
#undef MLHIP_CASE
    return grid_x;
}

}  // namespace mstats
}  // namespace mlhip
