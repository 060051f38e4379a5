// Dimensions beyond kMaxDim (d > 128): the register-resident kernels of this library keep a sample's coordinates (or a tile of
// them) in VGPRs / LDS and are instantiated per dimension up to 128. The reference has no such limit (ML/EM.cpp:96-101 only asks
// for d >= 1, N >= K), so above it the same three passes run here in a plain form -- one lane per sample (or per partial sum),
// coordinates re-read from memory (L1 / L2) at every use, parameters through scalar loads, no matrix cores. Same arithmetic and
// the same record / statistics layouts as the tuned kernels, so everything around them (closing arithmetic on the host,
// reductions, labels, initialisers, the C ABI) is unchanged. Correctness tier: O(d^2) loads per (sample, component); nothing
// here is tuned.
//   * em_estep_generic_kernel     EM::expectation_step        (ML/EM.cpp:190-219, ML/LinearAlgebra.cpp:8-31)
//   * em_mstats_generic_kernel    EM::maximisation_step sums  (ML/EM.cpp:229-248, ML/LinearAlgebra.cpp:54-73), sample covariance
//   * kmeans_assign_generic_kernel  KMeans::assignment_step   (ML/KMeans.cpp:153-178)
#include "device.hpp"
#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"

namespace mlhip {
namespace {

__device__ __forceinline__ double wave_sum_fixed(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/// Records: estep_param_stride(D) doubles per component, [ mean(D) | W = L^-1 packed lower triangle | coef ] (layout.hpp).
__global__ __launch_bounds__(256) void em_estep_generic_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n, uint32_t n_pad,
                                                                int D, const double* __restrict__ params, int K,
                                                                double* __restrict__ lw_out, size_t ldr, double* __restrict__ lse_out,
                                                                double* __restrict__ ll_partials)
{
    __shared__ double red[4];
    const size_t PS = (size_t)D + (size_t)D * (D + 1) / 2 + 1;
    double ll_acc = 0.0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_pad; i += gridDim.x * 256u) {
        const double* __restrict__ xi = xt + i;
        double m = -__builtin_inf(), s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double* __restrict__ p = params + (size_t)k * PS;        // wave-uniform: scalar loads
            const double* __restrict__ w = p + D;
            double q = 0.0;
            for (int j = 0; j < D; ++j) {
                const double* __restrict__ wj = w + (size_t)j * (j + 1) / 2;
                double y = wj[0] * (xi[0] - p[0]);                          // the order of em_estep_kernel: row j, ascending l
                for (int l = 1; l <= j; ++l) y = __builtin_fma(wj[l], xi[(size_t)l * ldx] - p[l], y);
                q = __builtin_fma(y, y, q);
            }
            const double lw = __builtin_fma(-0.5, q, p[PS - 1]);
            lw_out[(size_t)k * ldr + i] = lw;
            const double e = exp_nonpos(lw == -HUGE_VAL ? -HUGE_VAL : -fabs(lw - m));   // online log-sum-exp, one exp per component (lw = -inf: adds 0)
            const bool up = lw > m;
            s = up ? __builtin_fma(s, e, 1.0) : s + e;
            m = up ? lw : m;
        }
        const double lse = m + log(s);
        lse_out[i] = lse;
        if (i < n) ll_acc += lse;
    }
    ll_acc = wave_sum_fixed(ll_acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ll_acc;
    __syncthreads();
    if (threadIdx.x == 0) ll_partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

/// One workgroup per (row a of the packed lower triangle of xt xt^T, component k): entries (a, b), b <= a, in chunks of BT
/// columns; every thread sums its samples (stride 256, ascending), the 256 partial sums are combined in a fixed order. Writes
/// ONE partial block [K][F] (KP = K, FP = F), which launch_em_reduce folds into the statistics like any other.
constexpr int BT = 8;
__global__ __launch_bounds__(256) void em_mstats_generic_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n, int d,
                                                                 const double* __restrict__ shift, const double* __restrict__ lw,
                                                                 size_t ldr, const double* __restrict__ lse, int mode,
                                                                 double* __restrict__ partials, int F)
{
    __shared__ double red[4][BT];
    const int a = blockIdx.x, k = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* __restrict__ wk = lw + (size_t)k * ldr;
    const double sa = a < d ? shift[a] : 0.0;
    double* __restrict__ out = partials + (size_t)k * F + (size_t)a * (a + 1) / 2;
    for (int b0 = 0; b0 <= a; b0 += BT) {
        double acc[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) acc[t] = 0.0;
        for (uint32_t i = tid; i < n; i += 256u) {
            const double r = mode == kFromResp ? wk[i] : exp_nonpos(wk[i] - lse[i]);
            const double xa = a < d ? xt[(size_t)a * ldx + i] - sa : 1.0;
            const double w = r * xa;
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const int b = b0 + t;                                       // (wave-uniform)
                if (b <= a) {
                    const double xb = b < d ? xt[(size_t)b * ldx + i] - shift[b] : 1.0;
                    acc[t] = __builtin_fma(w, xb, acc[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const double v = wave_sum_fixed(acc[t]);
            if (lane == 0) red[wave][t] = v;
        }
        __syncthreads();
        if (tid < BT && b0 + tid <= a) out[b0 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        __syncthreads();
    }
}

/// One lane per sample, the reference's loop over the clusters (strict '<', ascending k) with the ascending-j fma chain of every
/// other kernel; [inertia, changed] per workgroup in the partial block's first two doubles. The update sums come from the
/// separate sweep (kmeans_update_kernel), which is dimension-generic.
__global__ __launch_bounds__(256) void kmeans_assign_generic_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n, int D,
                                                                     const double* __restrict__ cent, int K,
                                                                     uint32_t* __restrict__ labels, const uint32_t* __restrict__ old_labels,
                                                                     int have_old, double* __restrict__ min_dist,
                                                                     double* __restrict__ partials, size_t pstride)
{
    __shared__ double red[8];
    double inertia = 0.0, changed = 0.0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const double* __restrict__ xi = xt + i;
        double best = __builtin_inf();
        uint32_t arg = 0;
        for (int k = 0; k < K; ++k) {
            const double* __restrict__ c = cent + (size_t)k * D;            // wave-uniform: scalar loads
            double s = 0.0;
            for (int j = 0; j < D; ++j) {
                const double t = xi[(size_t)j * ldx] - c[j];
                s = __builtin_fma(t, t, s);
            }
            if (s < best) { best = s; arg = (uint32_t)k; }
        }
        labels[i] = arg;
        if (min_dist) min_dist[i] = best;
        inertia += best;
        changed += (!have_old || old_labels[i] != arg) ? 1.0 : 0.0;
    }
    inertia = wave_sum_fixed(inertia);
    changed = wave_sum_fixed(changed);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = inertia;
        red[4 + (threadIdx.x >> 6)] = changed;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* my_part = partials + (size_t)blockIdx.x * pstride;
        my_part[0] = ((red[0] + red[1]) + red[2]) + red[3];
        my_part[1] = ((red[4] + red[5]) + red[6]) + red[7];
    }
}

}  // namespace

int launch_em_estep_generic(const EstepArgs& a, hipStream_t stream)
{
    const uint32_t n_pad = padded_samples(a.n);
    const uint32_t blocks_needed = n_pad / 256;
    const int grid = (int)(blocks_needed < (uint32_t)a.n_ll_partials ? blocks_needed : (uint32_t)a.n_ll_partials);
    hipLaunchKernelGGL(em_estep_generic_kernel, dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, a.n, n_pad, a.D, a.params, a.K, a.lw,
                       a.ldr, a.lse, a.ll_partials);
    return grid;
}

int launch_em_mstats_generic(const MstatsArgs& a, hipStream_t stream)
{
    const int F = stats_count(a.d);
    if ((size_t)a.K * F > a.partials_capacity) return -2;
    if (a.mode == kFromLogRespSelfNorm) return -3;
    hipLaunchKernelGGL(em_mstats_generic_kernel, dim3(a.d + 1, a.K), dim3(256), 0, stream, a.xt, a.ldx, a.n, a.d, a.shift, a.lw, a.ldr,
                       a.lse, a.mode, a.partials, F);
    return 1;                                                               // one partial block
}

void launch_kmeans_assign_generic(const KmeansArgs& a, int grid, size_t pstride, hipStream_t stream)
{
    hipLaunchKernelGGL(kmeans_assign_generic_kernel, dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, a.n, a.D, a.centroids, a.K, a.labels,
                       a.old_labels, a.have_old, a.min_dist, a.partials, pstride);
}

}  // namespace mlhip
