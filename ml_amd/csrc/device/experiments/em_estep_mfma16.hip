// E-step of Gaussian-mixture EM on the gfx950 fp64 matrix cores (dimensions 12..32) -- replaces
// EM::expectation_step (reference ML/EM.cpp:190-219) and its xAx_symmetric calls (ML/LinearAlgebra.cpp:8-31).
//
// Same arithmetic as em_estep.hip (z = x - mu_k, y = W_k z with W_k = L_k^-1 lower triangular, q = |y|^2,
// lw = log pi_k - sum log L_jj - q/2, online log-sum-exp), but the whitening product runs as a block-triangular
// GEMM on v_mfma_f64_16x16x4_f64:
//
//     Y[16 rows j][16 samples] += W[16 rows j][4 cols l] * Z[4 rows l][16 samples]
//
// A = a 16x4 slab of W_k (one double per lane, pre-arranged by the host in lane order: one coalesced 512-B load per
// slab, shared through L1/L2 by every wave), B = z = x - mu_k with the samples' coordinates held in VGPRs in operand
// layout for the whole component loop (64 samples per wave, 32 doubles per lane at d = 32). Only the slabs on or
// below the block diagonal are issued: 12 MFMAs per 16 samples per component at d = 32 (3 of 4 16x16 blocks).
// Why MFMA although the reference loop is a per-sample quadratic form: on MI355X fp64 MFMA and fp64 VALU draw on
// the same throughput (tools/microbench_fp64: 72 + 0 or 0 + 62 TFLOP/s, 72.5 together), and the VALU form needs one
// wave-uniform operand per FMA, which the scalar path cannot deliver (SGPR spills, 2 waves/SIMD); the MFMA form
// reuses each W slab for 4 x 16 samples from registers.
//
// After the chain, lane (g = lane>>4, s = lane&15) holds y[16J + g + 4r][s] in register r of block J; the squares are
// summed per lane, across the 4 lane groups by two xor-shuffles, and lane (g, s) keeps the q of sample 16g + s, so the
// log-domain epilogue (one exp per component) and the LW store are one sample per lane, fully coalesced.
#include "../device.hpp"

namespace mlhip {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int D> struct Shape {
    static constexpr int LS = D / 4;                 // 4-column slabs of W
    static constexpr int JB = (D + 15) / 16;         // 16-row blocks of W
    static constexpr int slabs_of(int J) { return (4 * (J + 1) < LS) ? 4 * (J + 1) : LS; }
    static constexpr int NC = (JB == 1) ? slabs_of(0) : slabs_of(0) + slabs_of(1);
    static constexpr int PS = NC * 64 + D + 1;       // doubles per component record
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/// Sum over the 4 lane groups (g = lane>>4) of v[sb], delivered so that group g ends with the total of v[g]:
/// a reduce-scatter in two swap steps (rows g <-> g^1 with v_permlane16_swap, halves g <-> g^2 with
/// v_permlane32_swap) -- 6 VALU swaps and 3 adds instead of 16 ds_bpermute and 8 adds.
__device__ __forceinline__ double reduce_scatter_groups(double v0, double v1, double v2, double v3)
{
    auto swap16 = [](double& a, double& b) {   // odd rows of a <-> even rows of b
        const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    };
    auto swap32 = [](double& a, double& b) {   // upper half of a <-> lower half of b
        const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    };
    swap16(v0, v1);            // even rows: v0 own, v1 <- odd row's v0 ; odd rows: v0 <- even row's v1, v1 own
    swap16(v2, v3);
    double t01 = v0 + v1;      // even rows: sum of v0 over the pair, odd rows: sum of v1
    double t23 = v2 + v3;      // even rows: v2, odd rows: v3
    swap32(t01, t23);          // lower half: t23 <- upper's t01 ; upper half: t01 <- lower's t23
    return t01 + t23;          // g=0: v0, g=1: v1, g=2: v2, g=3: v3 totals
}

/// Slab schedule of one component, ordered by column slab ls so that z = x - mu is formed once per (ls, sample block):
/// step t -> (J, ls). For ls < slabs_of(0) both row blocks use the slab column, J = 0 first.
template <int D> struct Steps {
    using S = Shape<D>;
    static constexpr int both = S::JB == 2 ? S::slabs_of(0) : 0;   // column slabs used by two row blocks
    static constexpr int J(int t) { return S::JB == 1 ? 0 : (t < 2 * both ? (t & 1) : 1); }
    static constexpr int ls(int t) { return S::JB == 1 ? t : (t < 2 * both ? t / 2 : t - both); }
    /// index of the slab in the host-side record (J-major: all J = 0 slabs first)
    static constexpr int slab(int t) { return J(t) == 0 ? ls(t) : S::slabs_of(0) + ls(t); }
    static constexpr bool first_of_ls(int t) { return S::JB == 1 || t >= 2 * both || (t & 1) == 0; }
};

template <int D>
__global__ __launch_bounds__(256, 2) void em_estep_mfma_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n,
                                                                uint32_t n_groups, const double* __restrict__ params,
                                                                int K, double* __restrict__ lw_out, size_t ldr,
                                                                double* __restrict__ lse_out,
                                                                double* __restrict__ ll_partials)
{
    using S = Shape<D>;
    using T = Steps<D>;
    constexpr int LS = S::LS, JB = S::JB, NC = S::NC, PS = S::PS;
    // rolling prefetch window of W slabs; W divides NC so that the slot of a step is a compile-time constant
    constexpr int W = (NC % 4 == 0) ? 4 : (NC % 3 == 0) ? 3 : (NC % 5 == 0) ? 5 : NC;
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, s = lane & 15;
    double ll_acc = 0.0;

    const uint32_t waves_total = gridDim.x * 4;
    for (uint32_t grp = blockIdx.x * 4 + wave; grp < n_groups; grp += waves_total) {
        const uint32_t base = grp * 64;
        // coordinates in B-operand layout: xb[ls][sb] = x[dim 4ls + g][sample base + 16sb + s]
        double xb[LS][4];
#pragma unroll
        for (int ls = 0; ls < LS; ++ls)
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) xb[ls][sb] = xt[(size_t)(4 * ls + g) * ldx + base + 16 * sb + s];

        double m = -__builtin_inf(), ssum = 0.0;

        // Rolling window over the slab stream of all components: slot t % W holds the slab of step t (and the mean
        // entry of its column slab); it is refilled with step t + W -- possibly of the next component -- as soon as
        // its 4 MFMAs are issued, so W slabs (about W * 256 MFMA cycles) of loads are always in flight.
        double aw[W], muw[W];
        auto fetch = [&](int slot, int k, int t) {
            const double* __restrict__ rec = params + (size_t)k * PS;
            aw[slot] = rec[T::slab(t) * 64 + lane];
            muw[slot] = rec[NC * 64 + 4 * T::ls(t) + g];
        };
#pragma unroll
        for (int t = 0; t < W; ++t) fetch(t, 0, t);

        for (int k = 0; k < K; ++k) {
            const double coef = params[(size_t)k * PS + NC * 64 + D];
            const int kn = k + 1 < K ? k + 1 : k;       // the last component prefetches itself again (discarded)
            d4 acc[4][JB];
            double z[4];
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const double a = aw[t % W];
                if (T::first_of_ls(t)) {
#pragma unroll
                    for (int sb = 0; sb < 4; ++sb) z[sb] = xb[T::ls(t)][sb] - muw[t % W];
                }
                if (t + W < NC) fetch(t % W, k, t + W); else fetch(t % W, kn, t + W - NC);
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) {
                    // first slab of a row block starts its chain from zero
                    const bool first = (T::ls(t) == 0);
                    acc[sb][T::J(t)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, z[sb], first ? d4{0.0, 0.0, 0.0, 0.0} : acc[sb][T::J(t)], 0, 0, 0);
                }
                // Pin the slab-major order: without it the scheduler regroups the MFMAs into per-sample-block chains
                // (fewer live accumulators) and pays the MFMA->VALU hazard wait after each of the 8 chains.
                __builtin_amdgcn_sched_barrier(0);
            }
            double qs[4];
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                double t2 = 0.0;
#pragma unroll
                for (int J = 0; J < JB; ++J)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) t2 = __builtin_fma(acc[sb][J][rr], acc[sb][J][rr], t2);
                qs[sb] = t2;
            }
            const double q = reduce_scatter_groups(qs[0], qs[1], qs[2], qs[3]);   // lane (g, s) gets sample 16g + s
            const double lw = __builtin_fma(-0.5, q, coef);
            lw_out[(size_t)k * ldr + base + lane] = lw;
            const double e = exp(-fabs(lw - m));
            const bool up = lw > m;
            ssum = up ? __builtin_fma(ssum, e, 1.0) : ssum + e;
            m = up ? lw : m;
        }
        const double lse = m + log(ssum);
        lse_out[base + lane] = lse;
        if (base + lane < n) ll_acc += lse;
    }
    ll_acc = wave_sum(ll_acc);
    if (lane == 0) red[wave] = ll_acc;
    __syncthreads();
    if (threadIdx.x == 0) ll_partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <int D>
int launch_t(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t n_pad = padded_samples(a.n);
    const uint32_t n_groups = n_pad / 64;
    uint32_t grid = (n_groups + 3) / 4;
    const uint32_t cap = (uint32_t)num_cus * 2;          // 2 workgroups (8 waves) per CU, persistent
    if (grid > cap) grid = cap;
    if (grid > (uint32_t)a.n_ll_partials) grid = (uint32_t)a.n_ll_partials;
    hipLaunchKernelGGL(em_estep_mfma_kernel<D>, dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, a.n, n_groups, a.params, a.K,
                       a.lw, a.ldr, a.lse, a.ll_partials);
    return (int)grid;
}

}  // namespace

static_assert(Shape<32>::NC == 12 && Shape<20>::NC == 9 && Shape<16>::NC == 4 && Shape<12>::NC == 3, "slab count");

int launch_em_estep_mfma(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    switch (a.D) {
    case 12: return launch_t<12>(a, num_cus, stream);
    case 16: return launch_t<16>(a, num_cus, stream);
    case 20: return launch_t<20>(a, num_cus, stream);
    case 24: return launch_t<24>(a, num_cus, stream);
    case 28: return launch_t<28>(a, num_cus, stream);
    case 32: return launch_t<32>(a, num_cus, stream);
    default: return -1;
    }
}

}  // namespace mlhip
