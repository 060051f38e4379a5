// E-step of Gaussian-mixture EM for gfx950 -- replaces EM::expectation_step (reference ML/EM.cpp:190-219)
// and the xAx_symmetric calls inside it (ML/LinearAlgebra.cpp:8-31).
//
// One lane owns one sample: its D coordinates live in VGPRs for the whole component loop (X is stored
// dimension-major in HBM, so the D loads of a wave are 512-B coalesced rows). The per-component record
// (mean, W = L^-1 packed lower triangle, coef) is wave-uniform: the compiler fetches it with scalar loads
// and feeds v_fma_f64 from SGPR pairs, so no LDS and no vector-memory traffic is spent on parameters.
//
//   z = x - mu_k ;  y = W_k z  (triangular, D(D+1)/2 FMA) ;  q = |y|^2 = z^T Sigma_k^-1 z
//   lw_k = log pi_k - sum log L_jj - q/2            (log-domain form of ML/EM.cpp:207-209)
//   lse  = log sum_k exp(lw_k)   (online log-sum-exp) ;  log-likelihood partial = sum_i lse_i  (:211)
//
// The row normalisation r_ik = exp(lw_ik - lse_i) (:214-218) is applied by the consumers
// (em_mstats.hip, em_post.hip), so the N x K block is written once here and read once there.
#include "device.hpp"
#include "exp_nonpos.hpp"

namespace mlhip {
namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/// Sum over a 256-thread block; result valid in thread 0. Deterministic.
__device__ __forceinline__ double block_sum_256(double v, double* smem4)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem4[wave] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

template <int D>
__global__ __launch_bounds__(256) void em_estep_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n,
                                                        uint32_t n_pad, const double* __restrict__ params, int K,
                                                        double* __restrict__ lw_out, size_t ldr,
                                                        double* __restrict__ lse_out, double* __restrict__ ll_partials)
{
    constexpr int PS = D + D * (D + 1) / 2 + 1;
    __shared__ double red[4];
    double ll_acc = 0.0;

    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_pad; i += gridDim.x * 256u) {
        double x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = xt[(size_t)j * ldx + i];

        double m = -__builtin_inf(), s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double* __restrict__ p = params + (size_t)k * PS;   // wave-uniform -> scalar loads
            double z[D];
#pragma unroll
            for (int j = 0; j < D; ++j) z[j] = x[j] - p[j];
            const double* __restrict__ w = p + D;
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double y = w[j * (j + 1) / 2] * z[0];
#pragma unroll
                for (int l = 1; l <= j; ++l) y = __builtin_fma(w[j * (j + 1) / 2 + l], z[l], y);
                q = __builtin_fma(y, y, q);
                // Keeps the scalar loads of later rows from being hoisted (and spilled) above this point.
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const double lw = __builtin_fma(-0.5, q, p[PS - 1]);
            lw_out[(size_t)k * ldr + i] = lw;
            // online log-sum-exp with a single exp per component
            // (a component of mixing weight 0 has lw = -inf: it adds exp(-inf) = 0 like the reference's `column *= 0`, ML/EM.cpp:209 --
            // -inf - -inf would be a NaN while the running maximum is still -inf; a genuinely NaN lw stays a NaN)
            const double e = exp_nonpos(lw == -HUGE_VAL ? -HUGE_VAL : -fabs(lw - m));
            const bool up = lw > m;
            s = up ? __builtin_fma(s, e, 1.0) : s + e;
            m = up ? lw : m;
        }
        const double lse = m + log(s);
        lse_out[i] = lse;
        if (i < n) ll_acc += lse;
    }
    const double total = block_sum_256(ll_acc, red);
    if (threadIdx.x == 0) ll_partials[blockIdx.x] = total;
}

template <int D>
int launch_t(const EstepArgs& a, hipStream_t stream)
{
    const uint32_t n_pad = padded_samples(a.n);
    const uint32_t blocks_needed = n_pad / 256;
    const int grid = (int)(blocks_needed < (uint32_t)a.n_ll_partials ? blocks_needed : (uint32_t)a.n_ll_partials);
    hipLaunchKernelGGL(em_estep_kernel<D>, dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, a.n, n_pad, a.params, a.K,
                       a.lw, a.ldr, a.lse, a.ll_partials);
    return grid;
}

}  // namespace

int launch_em_estep(const EstepArgs& a, hipStream_t stream)
{
    switch (a.D) {
    case 1: return launch_t<1>(a, stream);
    case 2: return launch_t<2>(a, stream);
    case 3: return launch_t<3>(a, stream);
    case 4: return launch_t<4>(a, stream);
    case 6: return launch_t<6>(a, stream);
    case 8: return launch_t<8>(a, stream);
    case 12: return launch_t<12>(a, stream);
    case 16: return launch_t<16>(a, stream);
    case 20: return launch_t<20>(a, stream);
    case 24: return launch_t<24>(a, stream);
    case 28: return launch_t<28>(a, stream);
    case 32: return launch_t<32>(a, stream);
    default:
        if (a.D <= kMaxDim) return -1;
        if (big_dim_applies(a.D)) {
            int cus = a.num_cus;                              // (per context: a device group may span different devices -- ADVICE r4)
            if (cus < 1) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            }
            return launch_em_estep_big(a, cus, stream);
        }
        return launch_em_estep_generic(a, stream);
    }
}

}  // namespace mlhip
