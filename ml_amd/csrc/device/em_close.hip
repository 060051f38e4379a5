// Closing arithmetic of the M-step ON THE DEVICE -- what the reference does per component at the end of
// EM::maximisation_step and in EM::process_covariances (reference ML/EM.cpp:242, 250-257, 274-287): from the all-reduced
// sufficient statistics to the new mixing weight, mean and covariance, the Cholesky factor of the covariance, its inverse
// W = L^-1, sum log L_jj, and the packed parameter record of the next E-step. One workgroup per component; K of them run side
// by side. With it an EM iteration is a fixed sequence of launches with a 1 KB read-back (log-likelihood sum, refinement
// flags, the FOLD criterion) instead of 0.3 MB down, K factorizations on host threads and 0.3 MB up (mlhip_em_iterate).
//
// The arithmetic is the host's (host/em_math.cpp: finalize_mstep, cholesky_lower, whitening_matrix, the record builders),
// statement by statement and in the same order, compiled with contraction off: every thread evaluates the same sequential
// dot products the host loops evaluate, so the parameters agree with the host path bit for bit except through log()
// (one ulp of the library functions), and all ranks of a row-sharded job hold bit-identical parameters.
#include "device.hpp"
#include "em_close_body.hpp"

#pragma clang fp contract(off)     // the host's closing arithmetic, statement by statement (see above)

namespace mlhip {
namespace {

using closing::NT;

template <int LAYOUT, int DT>
__global__ __launch_bounds__(NT) void em_close_kernel(const double* __restrict__ stats, int K, int d, int D,
                                                        const double* __restrict__ shift, double n_global, double refine_limit,
                                                        double* __restrict__ mixing, double* __restrict__ means,
                                                        double* __restrict__ covs, double* __restrict__ records, int PS,
                                                        double* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    closing::close_component<LAYOUT, DT>(stats, K, d, D, shift, n_global, refine_limit, mixing, means, covs, records, PS, info,
                                         (int)blockIdx.x, (int)threadIdx.x, sm);
}

/// Diagonal covariances: elementwise (host/em_math.cpp finalize_mstep_diag + build_diag_params). One workgroup per
/// component (d <= 32 threads busy); records of the padding rows k >= K are written once by the host and left alone.
__global__ __launch_bounds__(64) void em_close_diag_kernel(const double* __restrict__ stats, int K, int KP, int d, int D,
                                                            const double* __restrict__ shift, double n_global, double refine_limit,
                                                            double* __restrict__ mixing, double* __restrict__ means,
                                                            double* __restrict__ vars, double* __restrict__ records,
                                                            double* __restrict__ info)
{
    __shared__ double logs[64];
    __shared__ int codes[64];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int F = 2 * d + 1, PS = diag_param_stride_c(D);
    const double* s = stats + (size_t)k * F;
    const double s0 = s[2 * d];
    const double mix = s0 / n_global;
    double* rec = records + (size_t)k * PS;
    double* aT = records + (size_t)KP * PS;                   // the trailer behind the KP records (layout.hpp)
    double* bT = aT + (size_t)D * KP;
    if (tid < d) {
        const double mm = s[tid] / s0;
        const double mean = shift[tid] + mm;
        const double var = (s[d + tid] - s[tid] * mm) / s0 + 1e-15;
        means[(size_t)k * d + tid] = mean;
        vars[(size_t)k * d + tid] = var;
        const double off = mean - shift[tid];
        codes[tid] = (!isfinite(off) || !isfinite(var)) ? 2 : ((refine_limit > 0 && off * off > refine_limit * var) ? 1 : 0);
        const double l = sqrt(var);
        rec[tid] = mean;
        rec[D + tid] = (1.0 / l) / l;
        logs[tid] = log(l);
        // operands of the two-operation density form (layout.hpp diag_param_stride; host: build_diag_params)
        const double a = 1.0 / l;
        const double b = -(off * a);
        aT[(size_t)tid * KP + k] = a;
        bT[(size_t)tid * KP + k] = b;
        logs[32 + tid] = isfinite(a) ? b : __builtin_inf();
    }
    __syncthreads();
    if (tid == 0) {
        mixing[k] = mix;
        double ldh = 0.0;
        for (int j = 0; j < d; ++j) ldh += logs[j];
        rec[2 * D] = log(mix) - ldh;
        double b2 = 0.0;                                       // B2 = sum_j b_j^2 (layout.hpp), ascending j like the host
        for (int a = 0; a < d; ++a) b2 = __builtin_fma(logs[32 + a], logs[32 + a], b2);
        rec[2 * D + 1] = b2;
        int flag = 0;
        if (mix > 0 && isfinite(mix))
            for (int a = 0; a < d; ++a) {
                if (codes[a] == 2) break;
                if (codes[a] == 1) { flag = 1; break; }
            }
        info[1 + k] = flag;
        info[1 + K + k] = 0.0;
        if (k == 0) info[0] = stats[(size_t)K * F];
    }
}

}  // namespace

size_t em_close_info_doubles(int K) { return 1 + 2 * (size_t)K; }
bool em_close_supported(int d) { return (d >= 1 && d <= kMidDim) || em_close_big_supported(d); }
size_t em_close_work_doubles(int d, int K) { return em_close_big_supported(d) ? em_close_big_work_doubles(d, K) : 0; }

void launch_em_close(const CloseArgs& a, hipStream_t stream)
{
    const int d = a.d;
    if (d > kMidDim) { launch_em_close_big(a, stream); return; }
    const size_t smem = sizeof(double) * closing::scratch_doubles(d);
#define MLHIP_CLOSE(LAYOUT, DT, PS) \
    hipLaunchKernelGGL((em_close_kernel<LAYOUT, DT>), dim3(a.K), dim3(NT), smem, stream, a.stats, a.K, d, a.D, a.shift, a.n_global, \
                       a.refine_limit, a.mixing, a.means, a.covs, a.records, PS, a.info)
    if (a.layout == 2) {
        const int PS = estep_mfma4_param_stride(a.D);
        switch (a.D) {
        case 12: MLHIP_CLOSE(2, 12, PS); break;
        case 16: MLHIP_CLOSE(2, 16, PS); break;
        case 20: MLHIP_CLOSE(2, 20, PS); break;
        case 24: MLHIP_CLOSE(2, 24, PS); break;
        case 28: MLHIP_CLOSE(2, 28, PS); break;
        case 32: MLHIP_CLOSE(2, 32, PS); break;
        default: MLHIP_CLOSE(2, 0, PS); break;
        }
    } else {
        const int PS = estep_param_stride(a.D);
        switch (a.D) {
        case 1: MLHIP_CLOSE(0, 1, PS); break;
        case 2: MLHIP_CLOSE(0, 2, PS); break;
        case 3: MLHIP_CLOSE(0, 3, PS); break;
        case 4: MLHIP_CLOSE(0, 4, PS); break;
        case 6: MLHIP_CLOSE(0, 6, PS); break;
        case 8: MLHIP_CLOSE(0, 8, PS); break;
        default: MLHIP_CLOSE(0, 0, PS); break;
        }
    }
#undef MLHIP_CLOSE
}

void launch_em_close_diag(const CloseArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(em_close_diag_kernel, dim3(a.K), dim3(64), 0, stream, a.stats, a.K, mstats::em_diag_partial_rows(a.K), a.d, a.D, a.shift, a.n_global,
                       a.refine_limit, a.mixing, a.means, a.covs, a.records, a.info);
}

}  // namespace mlhip
