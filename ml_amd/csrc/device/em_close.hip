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

#pragma clang fp contract(off)     // the host's closing arithmetic, statement by statement (see above)

namespace mlhip {
namespace {

__device__ __forceinline__ int sidx(int a, int b) { return a * (a + 1) / 2 + b; }   // stats_index

constexpr int NT = 64;    // ONE wave per component: the factorization is a chain of short dependent steps, and a wave-wide
                          // barrier costs next to nothing where a 4-wave workgroup barrier per step cost 40 of 63 us (d = 32)

/// LAYOUT: 0 = estep_param_stride records (VALU E-step / fused small kernel), 2 = estep_mfma4_param_stride records.
/// DT: the padded dimension when d <= 32 (the thread's column of W = L^-1 then lives in registers, loops fully unrolled),
/// 0 = any d <= 64 (that column goes through LDS).
template <int LAYOUT, int DT>
__global__ __launch_bounds__(NT) void em_close_kernel(const double* __restrict__ stats, int K, int d, int D,
                                                        const double* __restrict__ shift, double n_global, double refine_limit,
                                                        double* __restrict__ mixing, double* __restrict__ means,
                                                        double* __restrict__ covs, double* __restrict__ records, int PS,
                                                        double* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int F = (d + 1) * (d + 2) / 2;
    double* s = sm;                    // F statistics of this component
    double* A = s + F;                 // d x d column-major: covariance, overwritten by its Cholesky factor (lower)
    double* W = A + d * d;             // d x d column-major: L^-1 (lower)
    double* m = W + d * d;             // d: S1'/S0
    double* mean = m + d;              // d
    double* c = mean + d;              // d: W (mean - shift)
    double* tcol = c + d;              // d: column scratch of the factorization
    __shared__ double s_ljj, s_ldh, s_mix;
    __shared__ int codes[128];
    const int k = blockIdx.x, tid = threadIdx.x;

    for (int e = tid; e < F; e += NT) s[e] = stats[(size_t)k * F + e];
    __syncthreads();
    const double s0 = s[sidx(d, d)];
    if (tid < d) {
        m[tid] = s[sidx(d, tid)] / s0;
        mean[tid] = shift[tid] + m[tid];
        means[(size_t)k * d + tid] = mean[tid];
    }
    if (tid == 0) { s_mix = s0 / n_global; mixing[k] = s_mix; }                      // ML/EM.cpp:257
    __syncthreads();
    for (int e = tid; e < d * d; e += NT) {
        const int a = e % d, b = e / d;                                              // element (a, b), column-major
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        double v = (s[sidx(hi, lo)] - s[sidx(d, hi)] * m[lo]) / s0;
        if (a == b) v += 1e-15;                                                      // ML/EM.cpp:252
        A[e] = v;
        covs[(size_t)k * d * d + e] = v;
    }
    __syncthreads();
    // refinement criterion of the host path (mlhip_abi.cpp finalize_out): scanned in order, a non-finite entry ends the scan
    if (tid < d) {
        const double off = mean[tid] - shift[tid], var = A[tid * d + tid];
        codes[tid] = (!isfinite(off) || !isfinite(var)) ? 2 : ((refine_limit > 0 && off * off > refine_limit * var) ? 1 : 0);
    }
    __syncthreads();
    if (tid == 0) {
        int flag = 0;
        if (s_mix > 0 && isfinite(s_mix))
            for (int a = 0; a < d; ++a) {
                if (codes[a] == 2) break;
                if (codes[a] == 1) { flag = 1; break; }
            }
        info[1 + k] = flag;
    }

    // ---- Cholesky (host/em_math.cpp cholesky_lower) and W = L^-1 (whitening_matrix).
    if constexpr (DT > 0) {
        // d <= 32: thread i keeps ROW i of the factor in registers; what another thread's row contributes arrives through
        // v_readlane (wave-uniform lane index -> an SGPR pair, used directly as the multiplier). No LDS round trips and no
        // barriers inside the factorization: the chain of dependent steps is the arithmetic itself. Every thread forms the
        // very dot products of the host loops, term by term in their order.
        auto lane_value = [](double v, int lane) {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
        };
        double Li[DT];
#pragma unroll
        for (int c0 = 0; c0 < DT; ++c0) Li[c0] = (tid < d && c0 < d) ? A[c0 * d + tid] : 0.0;   // A(tid, c0)
#pragma unroll
        for (int jj = 0; jj < DT; ++jj) {
            if (jj < d) {                                                                // (uniform)
                double t = Li[jj];
#pragma unroll
                for (int l = 0; l < jj; ++l) t -= Li[l] * lane_value(Li[l], jj);         // L(i,l) * L(j,l)
                const double ljj = sqrt(lane_value(t, jj));
                Li[jj] = tid == jj ? ljj : t / ljj;                                      // rows above the diagonal: unused
            }
        }
        // W, one thread per column `col`: w[i] = ((i == col) - sum_{l<i} L(i,l) w[l]) / L(i,i). Entries above the diagonal are
        // exact zeros, so the host's sum over l = col .. i-1 may as well start at l = 0 (t - L * 0 == t): same bits.
        double w[DT];
        const int col = tid;
#pragma unroll
        for (int ii = 0; ii < DT; ++ii) {
            double t = (ii == col) ? 1.0 : 0.0;
#pragma unroll
            for (int l = 0; l < ii; ++l) t -= lane_value(Li[l], ii < d ? ii : 0) * w[l];
            w[ii] = (ii < col || ii >= d) ? 0.0 : t / lane_value(Li[ii], ii < d ? ii : 0);
        }
        if (tid < d) {
#pragma unroll
            for (int c0 = 0; c0 < DT; ++c0)
                if (c0 < d) {
                    A[c0 * d + tid] = Li[c0];                                            // L back to LDS (log det, generic readers)
                    W[col * d + c0] = w[c0];
                }
        }
        __syncthreads();
    } else {
        for (int jj = 0; jj < d; ++jj) {
            if (tid >= jj && tid < d) {
                double t = A[jj * d + tid];
                for (int l = 0; l < jj; ++l) t -= A[l * d + tid] * A[l * d + jj];
                tcol[tid] = t;
            }
            __syncthreads();
            if (tid == 0) s_ljj = sqrt(tcol[jj]);
            __syncthreads();
            if (tid >= jj && tid < d) A[jj * d + tid] = tid == jj ? s_ljj : tcol[tid] / s_ljj;
            __syncthreads();
        }
        if (tid < d) {
            const int col = tid;
            for (int ii = 0; ii < d; ++ii) {
                if (ii < col) { W[col * d + ii] = 0.0; continue; }
                double t = (ii == col) ? 1.0 : 0.0;
                for (int l = col; l < ii; ++l) t -= A[l * d + ii] * W[col * d + l];
                W[col * d + ii] = t / A[ii * d + ii];
            }
        }
    }
    if (tid == 0) {
        double ldh = 0.0;
        for (int j = 0; j < d; ++j) ldh += log(A[j * d + j]);
        s_ldh = ldh;
    }
    __syncthreads();
    if (tid < d) {
        double acc = 0.0;
        for (int col = 0; col <= tid; ++col) acc += W[col * d + tid] * (mean[col] - shift[col]);
        c[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        double biggest = 0.0;
        bool finite = true;
        for (int j = 0; j < d; ++j) {
            const double a = fabs(c[j]);
            if (a > biggest) biggest = a;
            finite = finite && isfinite(c[j]);
        }
        info[1 + K + k] = finite ? biggest : __builtin_inf();
        if (k == 0) info[0] = stats[(size_t)K * F];                                  // the log-likelihood sum rides along
    }
    // ---- the next E-step's record
    double* rec = records + (size_t)k * PS;
    const double coef = log(s_mix) - s_ldh;
    if constexpr (LAYOUT == 2) {
        const int Q = D / 4, NB = Q * (Q + 1) / 2;
        for (int e = tid; e < NB * 16; e += NT) {
            const int t = e / 16, kk = (e % 16) / 4, i = e % 4;
            int C = 0;
            while (C + 1 < Q && (C + 1) * Q - (C + 1) * C / 2 <= t) ++C;             // column-quad-major block order
            const int R = C + (t - (C * Q - C * (C - 1) / 2));
            const int row = 4 * R + i, col = 4 * C + kk;
            rec[e] = (row < d && col <= row) ? W[col * d + row] : 0.0;
        }
        for (int j = tid; j < D; j += NT) {
            rec[NB * 16 + j] = j < d ? mean[j] : 0.0;
            rec[NB * 16 + D + j] = j < d ? -c[j] : 0.0;
        }
        if (tid == 0) rec[NB * 16 + 2 * D] = coef;
    } else {
        for (int j = tid; j < D; j += NT) rec[j] = j < d ? mean[j] : 0.0;
        for (int e = tid; e < D * (D + 1) / 2; e += NT) {
            int j = 0;
            while ((j + 1) * (j + 2) / 2 <= e) ++j;                                  // packed lower triangle, row by row
            const int l = e - j * (j + 1) / 2;
            rec[D + e] = (j < d) ? W[l * d + j] : 0.0;
        }
        if (tid == 0) rec[PS - 1] = coef;
    }
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
bool em_close_supported(int d) { return d >= 1 && d <= kMidDim; }

void launch_em_close(const CloseArgs& a, hipStream_t stream)
{
    const int d = a.d;
    const size_t smem = sizeof(double) * ((size_t)stats_count(d) + 2 * (size_t)d * d + 4 * (size_t)d);
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
