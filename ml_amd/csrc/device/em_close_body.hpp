// The per-component closing arithmetic of the M-step as a device function, shared by em_close.hip (one wave per component, K
// workgroups side by side) and em_resident.hip (the device-resident loop of short fits: every workgroup closes every iteration
// itself). See em_close.hip for what it computes and why it mirrors the host's arithmetic statement by statement.
#pragma once
#include "device.hpp"
#include "lane_ops.hpp"

namespace mlhip {
namespace closing {

__device__ __forceinline__ int sidx(int a, int b) { return a * (a + 1) / 2 + b; }   // stats_index

constexpr int NT = 64;    // ONE wave per component: the factorization is a chain of short dependent steps, and a wave-wide
                          // barrier costs next to nothing where a 4-wave workgroup barrier per step cost 40 of 63 us (d = 32)

/// LAYOUT: 0 = estep_param_stride records (VALU E-step / fused small kernel), 2 = estep_mfma4_param_stride records.
/// DT: the padded dimension when d <= 32 (the thread's column of W = L^-1 then lives in registers, loops fully unrolled),
/// 0 = any d <= 64 (that column goes through LDS).
/// LDS doubles one component's closing needs (statistics, covariance / factor, W, four d-vectors, three scalars, d flags).
__host__ __device__ constexpr size_t scratch_doubles(int d) { return (size_t)(d + 1) * (d + 2) / 2 + 2 * (size_t)d * d + 4 * (size_t)d + 4 + (size_t)(d + 1) / 2 + 1; }

/// All NT threads of the ONE wave that closes a component meet here: LDS writes of the lanes before it are visible to the lanes
/// after it (a wave executes in lockstep; the fences keep the compiler and the LDS counter in line).
#define MLHIP_CLOSE_SYNC()                                          \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      \
    } while (0)

/// Component k, by the NT threads tid = 0 .. NT - 1 of one wave, with `sm` = scratch_doubles(d) doubles of LDS of their own.
/// STATS_LOCAL: `stats` is LDS of the calling workgroup, written behind a workgroup barrier (the device-resident loop): read in
/// place; otherwise (global memory: the closing kernel) the component's F sums are staged into LDS first.
/// The order of the steps (and which of them share a lane barrier) is chosen for latency; every VALUE is formed by the host's
/// operations on the host's operands, so the bits are the host's (host/em_math.cpp finalize_mstep).
template <int LAYOUT, int DT, bool STATS_LOCAL = false, typename Probe = NoProbe>
__device__ __forceinline__ void close_component(const double* __restrict__ stats, int K, int d, int D, const double* __restrict__ shift,
                                                double n_global, double refine_limit, double* __restrict__ mixing,
                                                double* __restrict__ means, double* __restrict__ covs, double* __restrict__ records,
                                                int PS, double* __restrict__ info, const int k, const int tid, double* sm,
                                                const Probe& probe = Probe())
{
#pragma clang fp contract(off)     // the host's closing arithmetic, statement by statement (em_close.hip)
    const int F = (d + 1) * (d + 2) / 2;
    double* staged = sm;               // F statistics of this component (unless STATS_LOCAL)
    double* A = staged + F;            // d x d column-major: covariance, overwritten by its Cholesky factor (lower)
    double* W = A + d * d;             // d x d column-major: L^-1 (lower)
    double* m = W + d * d;             // d: S1'/S0 (kept for readers of the scratch; the lanes below form the quotients they need themselves)
    double* mean = m + d;              // d
    double* c = mean + d;              // d: W (mean - shift)
    double* tcol = c + d;              // d: column scratch of the factorization, then the d logarithms
    double& s_ljj = tcol[d];
    double& s_ldh = tcol[d + 1];
    double& s_mix = tcol[d + 2];
    int* codes = reinterpret_cast<int*>(tcol + d + 4);     // d ints

    const double* s = stats + (size_t)k * F;
    if constexpr (!STATS_LOCAL) {
        for (int e = tid; e < F; e += NT) staged[e] = s[e];
        MLHIP_CLOSE_SYNC();
        s = staged;
    }
    const double s0 = s[sidx(d, d)];
    if (tid < d) {
        const double mt = s[sidx(d, tid)] / s0;
        m[tid] = mt;
        mean[tid] = shift[tid] + mt;
        means[(size_t)k * d + tid] = mean[tid];
    }
    const double mix = s0 / n_global;                                                // ML/EM.cpp:257 (every lane: the same value)
    if (tid == 0) { s_mix = mix; mixing[k] = mix; }
    for (int e = tid; e < d * d; e += NT) {
        const int a = e % d, b = e / d;                                              // element (a, b), column-major
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const double mlo = s[sidx(d, lo)] / s0;                                      // = m[lo]: formed again here instead of fetched through LDS behind a barrier
        double v = (s[sidx(hi, lo)] - s[sidx(d, hi)] * mlo) / s0;
        if (a == b) v += 1e-15;                                                      // ML/EM.cpp:252
        A[e] = v;
        covs[(size_t)k * d * d + e] = v;
    }
    MLHIP_CLOSE_SYNC();
    // refinement criterion of the host path (runtime/em.cpp finalize_out); the codes are scanned further down, behind the next barrier
    if (tid < d) {
        const double off = mean[tid] - shift[tid], var = A[tid * d + tid];
        codes[tid] = (!isfinite(off) || !isfinite(var)) ? 2 : ((refine_limit > 0 && off * off > refine_limit * var) ? 1 : 0);
    }
    double log_mix;                                                                  // log of the mixing weight: needed at the very end

    probe(12);
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
        log_mix = log(mix);                                                              // (independent work for the stalls of the chain below)
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
        probe(13);
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
        probe(14);
        if (tid < d) {
#pragma unroll
            for (int c0 = 0; c0 < DT; ++c0)
                if (c0 < d) {
                    A[c0 * d + tid] = Li[c0];                                            // L back to LDS (log det, generic readers)
                    W[col * d + c0] = w[c0];
                }
        }
        MLHIP_CLOSE_SYNC();
    } else {
        for (int jj = 0; jj < d; ++jj) {
            if (tid >= jj && tid < d) {
                double t = A[jj * d + tid];
                for (int l = 0; l < jj; ++l) t -= A[l * d + tid] * A[l * d + jj];
                tcol[tid] = t;
            }
            MLHIP_CLOSE_SYNC();
            if (tid == 0) s_ljj = sqrt(tcol[jj]);
            MLHIP_CLOSE_SYNC();
            if (tid >= jj && tid < d) A[jj * d + tid] = tid == jj ? s_ljj : tcol[tid] / s_ljj;
            MLHIP_CLOSE_SYNC();
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
        log_mix = log(mix);
    }
    probe(15);
    // sum_j log L_jj in the host's order (ascending j, one addition at a time) -- the d logarithms themselves side by side, one per
    // lane: evaluated one after the other by a single lane they were a third of this function's time (d = 4: 1.2 of 4.8 us; d = 32:
    // 32 dependent calls of ~0.25 us each)
    for (int j = tid; j < d; j += NT) tcol[j] = log(A[j * d + j]);
    MLHIP_CLOSE_SYNC();                                                              // (W, the logarithms and the codes: visible to every lane)
    probe(16);
    if (tid == 0) {
        // scanned in order, a non-finite entry ends the scan
        int flag = 0;
        if (s_mix > 0 && isfinite(s_mix))
            for (int a = 0; a < d; ++a) {
                if (codes[a] == 2) break;
                if (codes[a] == 1) { flag = 1; break; }
            }
        info[1 + k] = flag;
        double ldh = 0.0;
        for (int j = 0; j < d; ++j) ldh += tcol[j];
        s_ldh = ldh;
    }
    {
        // c = W (mean - shift) and max_j |c_j| (infinite when an entry is not finite): the maximum over the lanes on the vector unit
        double reach = 0.0;
        if (tid < d) {
            double acc = 0.0;
            for (int col = 0; col <= tid; ++col) acc += W[col * d + tid] * (mean[col] - shift[col]);
            c[tid] = acc;
            reach = isfinite(acc) ? fabs(acc) : __builtin_inf();
        }
        reach = wave_max_nonneg(reach);
        if (tid == 0) {
            info[1 + K + k] = reach;
            if (k == 0) info[0] = stats[(size_t)K * F];                              // the log-likelihood sum rides along
        }
    }
    MLHIP_CLOSE_SYNC();
    probe(17);
    // ---- the next E-step's record
    double* rec = records + (size_t)k * PS;
    const double coef = log_mix - s_ldh;
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

#undef MLHIP_CLOSE_SYNC

}  // namespace closing
}  // namespace mlhip
