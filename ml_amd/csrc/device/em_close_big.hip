// Closing arithmetic of the M-step on the device for 64 < d <= 1024 -- what em_close.hip does for d <= 64 with one wave per component
// and both d x d matrices in LDS (reference ML/EM.cpp:242, 250-257, 274-287: new mixing weight, mean, covariance, its Cholesky factor L,
// W = L^-1, sum log L_jj, the next E-step's record). Here the matrices live in global memory (a component's pair is 16 MB at d = 1024;
// they stay in L2) and the factorization runs as a short sequence of launches over column PANELS of 32:
//
//   prepare      statistics -> mean, covariance (-> the caller's pack and the workspace), refinement codes; W starts as the identity
//   per panel    chol_diag  : ONE wave per component factors the 32 x 32 diagonal block in registers (lane = row, partners by v_readlane)
//   (of L)       chol_rows  : every row below the block, one thread per row: the panel's own columns against the diagonal block
//                chol_trail : RIGHT-looking -- the panel's 32 terms go to every entry of the trailing matrix at once (one thread per
//                             row and 32 columns, the partner rows' values by scalar loads): all the parallelism of the factorization
//   per panel    whiten_solve : the panel's 32 rows of W = L^-1 against the diagonal block, one thread per column of W
//   (of W)       whiten_trail : their terms to every later row at once
//                (the same panel of L and of W in the same launches: panel_solve_kernel = chol_rows + whiten_solve,
//                 panel_trail_kernel = chol_trail + whiten_trail + the next panel's chol_diag -- 2 launches per panel)
//   finish       sum log L_jj, c = W (mean - shift), flags, the record in the E-step's layout, the info block
// A first form walked the earlier columns per panel (left-looking: one launch per panel and long chains per thread: 25 ms at d = 1024,
// K = 4); with the trailing updates every launch is short and wide (3 + 2 launches per panel).
//
// Until round 4 these dimensions closed on the host (K factorizations on a thread team: 1.7 ms of a 24 ms iteration at d = 128,
// K = 32; 60 of 70 ms at d = 1024, K = 4).
//
// Every VALUE is formed by the host's operations on the host's operands in the host's order (host/em_math.cpp finalize_mstep,
// cholesky_lower, whitening_matrix, the record builders): an entry of L or W receives its terms in ascending l, one product and one
// subtraction at a time (contraction off), then one division -- whichever thread evaluates it and however the loops are blocked. So
// the parameters agree with the host path bit for bit except through log() (one ulp of the library function), exactly as for d <= 64,
// and all ranks of a row-sharded job hold bit-identical parameters.
#include "device.hpp"

#pragma clang fp contract(off)     // the host's closing arithmetic, statement by statement (see above)

namespace mlhip {
namespace {

constexpr int PB = 32;             // panel width (columns of L / rows of W handled per step). 64: measured slower (closing 4.49 against 3.56 ms at
                                   // d = 1024, K = 4; 0.73 / 0.57 at d = 256): the in-panel chains grow faster than the launches shrink

__device__ __forceinline__ int sidx(int a, int b) { return a * (a + 1) / 2 + b; }   // stats_index

/// Workspace of one component (doubles): L (d x d column-major: L[l * d + i] = L(i, l); starts as the covariance), Wt (d x d:
/// Wt[i * d + c] = W(i, c)), mean (d), c = W (mean - shift) (d), log L_jj (d), refinement codes (d).
__host__ __device__ inline size_t big_stride(int d) { return 2 * (size_t)d * d + 4 * (size_t)d; }

struct BigView {
    double *L, *Wt, *mean, *c, *logs, *codes;
    __device__ BigView(double* work, int k, int d)
    {
        L = work + (size_t)k * big_stride(d);
        Wt = L + (size_t)d * d;
        mean = Wt + (size_t)d * d;
        c = mean + d;
        logs = c + d;
        codes = logs + d;
    }
};

__device__ __forceinline__ double lane_value(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

/// Covariance entries (one thread each), means, mixing weight, refinement codes (em_close_body.hpp's first part, from global memory).
__global__ __launch_bounds__(256) void close_big_prepare_kernel(const double* __restrict__ stats, int K, int d, const double* __restrict__ shift,
                                                                 double n_global, double refine_limit, double* __restrict__ mixing,
                                                                 double* __restrict__ means, double* __restrict__ covs, double* __restrict__ work)
{
    const int k = blockIdx.x;
    const int F = (d + 1) * (d + 2) / 2;
    const double* __restrict__ s = stats + (size_t)k * F;
    BigView v(work, k, d);
    const double s0 = s[sidx(d, d)];
    const size_t e = (size_t)blockIdx.y * 256 + threadIdx.x;
    if (e < (size_t)d * d) {
        const int a = (int)(e % d), b = (int)(e / d);                                // element (a, b), column-major
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const double mlo = s[sidx(d, lo)] / s0;
        double x = (s[sidx(hi, lo)] - s[sidx(d, hi)] * mlo) / s0;
        if (a == b) x += 1e-15;                                                      // ML/EM.cpp:252
        v.L[e] = x;
        v.Wt[e] = a == b ? 1.0 : 0.0;
        covs[(size_t)k * d * d + e] = x;
        if (a == b) {
            const double mean = shift[a] + mlo;
            const double off = mean - shift[a];
            v.mean[a] = mean;
            means[(size_t)k * d + a] = mean;
            v.codes[a] = (!isfinite(off) || !isfinite(x)) ? 2.0 : ((refine_limit > 0 && off * off > refine_limit * x) ? 1.0 : 0.0);
        }
    }
    if (e == 0) mixing[k] = s0 / n_global;                                           // ML/EM.cpp:257
}

/// The same start from GIVEN parameters (the records of a fit's first E-step: EM::fit's initial covariances, ML/EM.cpp:143-150 --
/// host/em_math.cpp build_estep_params*): the covariance goes into the workspace as it is, no statistics, no refinement codes.
__global__ __launch_bounds__(256) void close_big_from_params_kernel(const double* __restrict__ means, const double* __restrict__ covs, int d,
                                                                     double* __restrict__ work)
{
    const int k = blockIdx.x;
    BigView v(work, k, d);
    const size_t e = (size_t)blockIdx.y * 256 + threadIdx.x;
    if (e < (size_t)d * d) {
        const int a = (int)(e % d), b = (int)(e / d);
        v.L[e] = covs[(size_t)k * d * d + e];
        v.Wt[e] = a == b ? 1.0 : 0.0;
    }
    if (e < (size_t)d) {
        v.mean[e] = means[(size_t)k * d + e];
        v.codes[e] = 0.0;
    }
}

/// A 32 x 32 diagonal block held one row per lane (lanes 0 .. 31: t[c] = entry (row, column c) with the terms of all earlier panels in),
/// factored in registers as em_close_body.hpp factors a whole d <= 32 matrix.
/// Column by column, each finished column's term going to every later column at once (the 31 - l updates of a step are independent
/// of each other: issued back to back; walking an entry's terms only when its column comes up -- the form of em_close_body.hpp -- is
/// a dependent chain of up to 31 products per step). An entry still receives its terms in ascending l.
__device__ __forceinline__ void factor_diag_block(double (&t)[PB], int r, int nb)
{
#pragma unroll
    for (int l = 0; l < PB; ++l) {
        if (l < nb) {                                                                // (uniform)
            const double ljj = sqrt(lane_value(t[l], l));
            t[l] = r == l ? ljj : t[l] / ljj;                                        // (rows above the diagonal: unused)
#pragma unroll
            for (int c = l + 1; c < PB; ++c) t[c] -= t[l] * lane_value(t[l], c);     // L(i, j0 + l) * L(j0 + c, j0 + l)
        }
    }
}

/// The FIRST panel's diagonal block (rows 0 .. 31, lane = row), one wave per component. The later panels' blocks are factored by the
/// workgroup of the trailing update that has just put the last terms into them (chol_trail_body).
__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ work, int d, int j0)
{
    const int k = blockIdx.x, r = threadIdx.x;
    BigView v(work, k, d);
    const int nb = d - j0 < PB ? d - j0 : PB;
    const bool row = r < nb;
    const int i = j0 + (row ? r : 0);
    double t[PB];
#pragma unroll
    for (int c = 0; c < PB; ++c) t[c] = (row && c < nb) ? v.L[(size_t)(j0 + c) * d + i] : 0.0;
    factor_diag_block(t, r, nb);
    if (row) {
#pragma unroll
        for (int c = 0; c < PB; ++c)
            if (c <= r && c < nb) v.L[(size_t)(j0 + c) * d + i] = t[c];
    }
}

/// Rows below the panel's diagonal block, one thread per row: the panel's own columns against the diagonal block (final:
/// chol_diag_kernel ran before). `work_ro` is the same workspace, read-only here for everything this kernel reads through it (the
/// diagonal rows: other rows than the ones it writes) -- uniform addresses, scalar loads.
__device__ __forceinline__ void chol_rows_body(double* __restrict__ work, const double* __restrict__ work_ro, int d, int j0, int k, int by,
                                               double* __restrict__ Ld)
{
    BigView v(work, k, d);
    const double* __restrict__ Lro = work_ro + (size_t)k * big_stride(d);
    const int nb = d - j0 < PB ? d - j0 : PB;
    const int i_raw = j0 + PB + by * 64 + threadIdx.x;
    const bool row = i_raw < d;
    const int i = row ? i_raw : d - 1;
    // the 32 x 32 diagonal block -> LDS, column by column (coalesced); read back as broadcasts. (Scalar loads of the partner values, a
    // batch per column, were 32 dependent round trips of ~0.4 us: most of this kernel's 19 us.)
    for (int e = threadIdx.x; e < PB * PB; e += 64) {
        const int c = e / PB, c2 = e - c * PB;
        Ld[e] = (c < nb && c2 < nb) ? Lro[(size_t)(j0 + c) * d + j0 + c2] : 1.0;
    }
    double t[PB];
#pragma unroll
    for (int c = 0; c < PB; ++c) t[c] = c < nb ? v.L[(size_t)(j0 + c) * d + i] : 0.0;
    __syncthreads();
    // (column by column of the diagonal block: entry c is final once the terms of the columns before it are in, and its own term goes
    // to every later entry at once -- each entry still receives its terms in ascending order)
#pragma unroll
    for (int c = 0; c < PB; ++c) {
        if (c < nb) {                                                                // (uniform)
            const double* lc = Ld + c * PB;                                          // column j0 + c of the diagonal block
            t[c] = t[c] / lc[c];
#pragma unroll
            for (int c2 = c + 1; c2 < PB; ++c2) t[c2] -= t[c] * lc[c2];
        }
    }
    if (row) {
#pragma unroll
        for (int c = 0; c < PB; ++c)
            if (c < nb) v.L[(size_t)(j0 + c) * d + i] = t[c];
    }
}

/// Right-looking step of the factorization: the terms l = j0 .. j0 + 31 of the panel just finished go to the entries (i, jb + c),
/// c = 0 .. 31, of one later column block -- one thread per row i >= jb, 32 accumulators, the partner values L(jb + c, l) by scalar
/// loads. Every entry receives its terms in ascending l: panels in order, l in order inside a panel.
__device__ __forceinline__ void chol_trail_body(double* __restrict__ work, const double* __restrict__ work_ro, int d, int j0, int k, int by, int bz)
{
    BigView v(work, k, d);
    const double* __restrict__ Lro = work_ro + (size_t)k * big_stride(d);
    const int jb = j0 + PB + bz * PB;                                   // first column of the block
    const int nbc = d - jb < PB ? d - jb : PB;
    const int i_raw = jb + by * 64 + threadIdx.x;
    if (jb + by * 64 >= d) return;                                       // (uniform: no row of this block exists)
    const bool row = i_raw < d;
    const int i = row ? i_raw : d - 1;
    double t[PB];
#pragma unroll
    for (int c = 0; c < PB; ++c) t[c] = c < nbc ? v.L[(size_t)(jb + c) * d + i] : 0.0;
#pragma unroll 4
    for (int l = j0; l < j0 + PB; ++l) {
        const double xl = v.L[(size_t)l * d + i];                                    // L(i, l)
        const double* __restrict__ cl = Lro + (size_t)l * d + jb;                    // L(jb + c, l), c = 0 .. 31 (uniform)
#pragma unroll
        for (int c = 0; c < PB; ++c) t[c] -= xl * cl[c];
    }
    if (by == 0 && bz == 0) {
        // (uniform) this wave's lanes 0 .. 31 hold the NEXT panel's diagonal block, complete: factored here, behind the update, instead of
        // in a launch of its own between this one and the next panel's rows (the entries above the diagonal keep the update's values,
        // as that launch left them; lanes 32 .. 63 -- rows below the block -- are not touched: EXEC)
        const int r = threadIdx.x;
        if (row && r < PB) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (c < nbc && c > r) v.L[(size_t)(jb + c) * d + i] = t[c];
        }
        if (r < PB) factor_diag_block(t, r, nbc);
        if (row) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (c < nbc && (r >= PB || c <= r)) v.L[(size_t)(jb + c) * d + i] = t[c];
        }
    } else if (row) {
#pragma unroll
        for (int c = 0; c < PB; ++c)
            if (c < nbc) v.L[(size_t)(jb + c) * d + i] = t[c];
    }
}

/// W = L^-1 (host/em_math.cpp whitening_matrix): entry (i, c) starts from (i == c), receives L(i, l) W(l, c) for l = c .. i - 1 in
/// ascending order and is divided by L(i, i); for l < c the factor W(l, c) is an exact zero (stored as such: t - L * 0 == t), and the
/// entries above the diagonal are set to zero as the host sets them. Stored transposed (Wt[i * d + c]: the threads of a wave -- one
/// per column c -- touch neighbouring words). This kernel: the 32 rows i0 .. of one panel against the diagonal block of L, the terms of
/// all rows l < i0 being in (whiten_trail_kernel).
__device__ __forceinline__ void whiten_solve_body(double* __restrict__ work, const double* __restrict__ work_ro, int d, int i0, int k, int by,
                                                  double* __restrict__ Ld)
{
    BigView v(work, k, d);
    const double* __restrict__ Lro = work_ro + (size_t)k * big_stride(d);
    const int nb = d - i0 < PB ? d - i0 : PB;
    const int c_raw = by * 64 + threadIdx.x;
    const bool col = c_raw < i0 + nb;                                                // (columns right of the panel: zeros already)
    const int c = col ? c_raw : 0;
    // Ld: the diagonal block of L (as in chol_rows_body)
    for (int e = threadIdx.x; e < PB * PB; e += 64) {
        const int r = e / PB, r2 = e - r * PB;
        Ld[e] = (r < nb && r2 < nb) ? Lro[(size_t)(i0 + r) * d + i0 + r2] : 1.0;
    }
    double acc[PB];
#pragma unroll
    for (int r = 0; r < PB; ++r) acc[r] = r < nb ? v.Wt[(size_t)(i0 + r) * d + c] : 0.0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PB; ++r) {                                                   // (column by column of the diagonal block, as chol_rows_kernel)
        if (r < nb) {                                                                // (uniform)
            const int i = i0 + r;
            const double* lc = Ld + r * PB;                                          // column i of L, rows i0 ..
            acc[r] = i < c ? 0.0 : acc[r] / lc[r];
#pragma unroll
            for (int r2 = r + 1; r2 < PB; ++r2) acc[r2] -= lc[r2] * acc[r];
        }
    }
    if (col) {
#pragma unroll
        for (int r = 0; r < PB; ++r)
            if (r < nb) v.Wt[(size_t)(i0 + r) * d + c] = acc[r];
    }
}

/// The finished rows l = i0 .. i0 + 31 of W go to the 32 later rows ib .. of one row block: (i, c) -= L(i, l) W(l, c), ascending l.
/// Only the columns c < i0 + 32 can hold anything but zero.
__device__ __forceinline__ void whiten_trail_body(double* __restrict__ work, const double* __restrict__ work_ro, int d, int i0, int k, int by, int bz)
{
    BigView v(work, k, d);
    const double* __restrict__ Lro = work_ro + (size_t)k * big_stride(d);
    const int ib = i0 + PB + bz * PB;                                   // first row of the block
    const int nbr = d - ib < PB ? d - ib : PB;
    const int c_raw = by * 64 + threadIdx.x;
    const bool col = c_raw < i0 + PB;
    const int c = col ? c_raw : 0;
    double acc[PB];
#pragma unroll
    for (int r = 0; r < PB; ++r) acc[r] = r < nbr ? v.Wt[(size_t)(ib + r) * d + c] : 0.0;
#pragma unroll 4
    for (int l = i0; l < i0 + PB; ++l) {
        const double wl = v.Wt[(size_t)l * d + c];                                   // W(l, c)
        const double* __restrict__ cl = Lro + (size_t)l * d + ib;                    // L(ib + r, l), r = 0 .. 31 (uniform)
#pragma unroll
        for (int r = 0; r < PB; ++r) acc[r] -= cl[r] * wl;
    }
    if (col) {
#pragma unroll
        for (int r = 0; r < PB; ++r)
            if (r < nbr) v.Wt[(size_t)(ib + r) * d + c] = acc[r];
    }
}

/// Panel p of L and panel p of W = L^-1 advance TOGETHER (late round 5; before: all panels of L, then all panels of W -- 3 + 2 launches
/// of 14 - 18 us per panel, each a short chain of dependent steps in every thread, 160 of them in a row at d = 1024). The rows of W in
/// panel p need the diagonal block of L (chol_diag of the panel) and the terms of W's earlier panels; their terms for the later rows
/// need L's rows below the block in the panel's columns (chol_rows of the panel) -- nothing of L's LATER panels. So one launch carries
/// chol_rows and whiten_solve of a panel (workgroups blockIdx.y < ny_chol: rows of L; the others: columns of W), the next one
/// chol_trail and whiten_trail: 3 launches per panel, the same values by the same operations.
__global__ __launch_bounds__(64) void panel_solve_kernel(double* __restrict__ work, const double* __restrict__ work_ro, int d, int j0, int ny_chol)
{
    __shared__ double Ld[PB * PB];
    const int by = (int)blockIdx.y;
    if (by < ny_chol) chol_rows_body(work, work_ro, d, j0, (int)blockIdx.x, by, Ld);
    else whiten_solve_body(work, work_ro, d, j0, (int)blockIdx.x, by - ny_chol, Ld);
}

__global__ __launch_bounds__(64) void panel_trail_kernel(double* __restrict__ work, const double* __restrict__ work_ro, int d, int j0, int ny_chol)
{
    const int by = (int)blockIdx.y;
    if (by < ny_chol) chol_trail_body(work, work_ro, d, j0, (int)blockIdx.x, by, (int)blockIdx.z);
    else whiten_trail_body(work, work_ro, d, j0, (int)blockIdx.x, by - ny_chol, (int)blockIdx.z);
}

/// c = W (mean - shift): row i's sum over col = 0 .. i in ascending order, one product and one addition at a time (the host's loop).
/// One thread per row (64 of the workgroup's 256; all four waves load); the rows of W are read as 64 x 64 tiles through LDS (a thread walking its own row of Wt directly touches one
/// cache line per load and lane: 1.25 ms at d = 1024).
__global__ __launch_bounds__(256) void close_big_cvec_kernel(const double* __restrict__ shift, double* __restrict__ work, int d)
{
    __shared__ double tile[64][65];
    __shared__ double diff[64];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    BigView v(work, k, d);
    const int r0 = blockIdx.y * 64;
    const int i = r0 + tid;                                                          // (threads 0 .. 63 own a row)
    double acc = 0.0;
    for (int c0 = 0; c0 <= r0; c0 += 64) {
        __syncthreads();
        const int col = c0 + lane;
        if (wave == 0) diff[lane] = col < d ? v.mean[col] - shift[col] : 0.0;
        for (int r = wave; r < 64; r += 4) tile[r][lane] = (r0 + r < d && col < d) ? v.Wt[(size_t)(r0 + r) * d + col] : 0.0;
        __syncthreads();
        if (tid < 64) {
            const int n_cols = i - c0 + 1 < 64 ? i - c0 + 1 : 64;                    // columns c0 .. min(i, c0 + 63)
            for (int cc = 0; cc < n_cols; ++cc) acc += tile[tid][cc] * diff[cc];
        }
    }
    if (tid < 64 && i < d) v.c[i] = acc;
}

/// sum log L_jj, c = W (mean - shift), the refinement flag, the next E-step's record, the info block (em_close_body.hpp's last part,
/// from global memory). One workgroup per component.
template <int LAYOUT>
__global__ __launch_bounds__(256) void close_big_finish_kernel(const double* __restrict__ stats, int K, int d, int D,
                                                                const double* __restrict__ shift, const double* __restrict__ mixing,
                                                                double* __restrict__ work, double* __restrict__ records, int PS,
                                                                double* __restrict__ info)
{
    __shared__ double red[256];
    __shared__ double s_coef;
    __shared__ double s_logs[1024], s_codes[1024];                                   // (d <= 1024: thread 0's two ordered scans walk LDS, not
    const int k = blockIdx.x, tid = threadIdx.x;                                     //  1 024 dependent global loads each)
    const int F = (d + 1) * (d + 2) / 2;
    BigView v(work, k, d);
    if constexpr (LAYOUT == 0) {
        // The packed triangle of W -- D (D + 1) / 2 doubles, 4 MB at d = 1024 -- is copied by the workgroups blockIdx.y >= 1, rows dealt
        // round-robin (one workgroup moving all of it took 0.7 ms of the 0.8 this kernel ran at d = 1024); workgroup 0 does the rest.
        if (blockIdx.y > 0) {
            double* __restrict__ rec = records + (size_t)k * PS;
            const int wave = tid >> 6, lane = tid & 63;
            const int stride = 4 * ((int)gridDim.y - 1);
            for (int j = ((int)blockIdx.y - 1) * 4 + wave; j < D; j += stride) {
                double* __restrict__ out = rec + D + (size_t)j * (j + 1) / 2;
                for (int l = lane; l <= j; l += 64) out[l] = j < d ? v.Wt[(size_t)j * d + l] : 0.0;
            }
            return;
        }
    }
    const double mix = mixing[k];
    for (int j = tid; j < d; j += 256) {
        s_logs[j] = log(v.L[(size_t)j * d + j]);
        s_codes[j] = v.codes[j];
    }
    double reach = 0.0;
    for (int i = tid; i < d; i += 256) {
        const double acc = v.c[i];                                                   // (close_big_cvec_kernel)
        const double m = isfinite(acc) ? fabs(acc) : __builtin_inf();
        reach = m > reach ? m : reach;
    }
    red[tid] = reach;
    __syncthreads();
    if (tid == 0) {
        int flag = 0;                                                                // scanned in order, a non-finite entry ends the scan
        if (mix > 0 && isfinite(mix))
            for (int a = 0; a < d; ++a) {
                if (s_codes[a] == 2.0) break;
                if (s_codes[a] == 1.0) { flag = 1; break; }
            }
        info[1 + k] = flag;
        double ldh = 0.0;
        for (int j = 0; j < d; ++j) ldh += s_logs[j];                                // the host's order
        s_coef = log(mix) - ldh;
        double m = 0.0;
        for (int t = 0; t < 256; ++t) m = red[t] > m ? red[t] : m;
        info[1 + K + k] = m;
        if (k == 0 && stats) info[0] = stats[(size_t)K * F];                         // the log-likelihood sum rides along
    }
    __syncthreads();
    double* __restrict__ rec = records + (size_t)k * PS;
    if constexpr (LAYOUT == 2) {
        const int Q = D / 4, NB = Q * (Q + 1) / 2;
        for (int e = tid; e < NB * 16; e += 256) {
            const int t = e / 16, kk = (e % 16) / 4, i = e % 4;
            int C = 0;
            while (C + 1 < Q && (C + 1) * Q - (C + 1) * C / 2 <= t) ++C;             // column-quad-major block order
            const int R = C + (t - (C * Q - C * (C - 1) / 2));
            const int row = 4 * R + i, colm = 4 * C + kk;
            rec[e] = (row < d && colm <= row) ? v.Wt[(size_t)row * d + colm] : 0.0;
        }
        for (int j = tid; j < D; j += 256) {
            rec[NB * 16 + j] = j < d ? v.mean[j] : 0.0;
            rec[NB * 16 + D + j] = j < d ? -v.c[j] : 0.0;
        }
        if (tid == 0) rec[NB * 16 + 2 * D] = s_coef;
    } else {
        for (int j = tid; j < D; j += 256) rec[j] = j < d ? v.mean[j] : 0.0;
        if (tid == 0) rec[PS - 1] = s_coef;                                          // (the packed triangle: the workgroups blockIdx.y >= 1)
    }
}

/// L and W = L^-1 of the K matrices in the workspace, then the records / info block.
void factor_and_finish(const double* stats, const double* mixing, const CloseArgs& a, hipStream_t stream)
{
    const int d = a.d, K = a.K;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(K), dim3(64), 0, stream, a.work, d, 0);
    for (int j0 = 0; j0 < d; j0 += PB) {
        const int below = d - (j0 + PB);                                             // rows / columns behind the panel
        const int cols = j0 + PB < d ? j0 + PB : d;                                  // columns of W that hold anything but zero so far
        const int ny_chol = below > 0 ? (below + 63) / 64 : 0, ny_w = (cols + 63) / 64;
        hipLaunchKernelGGL(panel_solve_kernel, dim3(K, ny_chol + ny_w), dim3(64), 0, stream, a.work, a.work, d, j0, ny_chol);
        if (below > 0)
            hipLaunchKernelGGL(panel_trail_kernel, dim3(K, ny_chol + ny_w, (below + PB - 1) / PB), dim3(64), 0, stream, a.work, a.work, d, j0, ny_chol);
    }
    hipLaunchKernelGGL(close_big_cvec_kernel, dim3(K, (d + 63) / 64), dim3(256), 0, stream, a.shift, a.work, d);
    if (a.layout == 2) {
        const int PS = estep_mfma4_param_stride(a.D);
        hipLaunchKernelGGL(close_big_finish_kernel<2>, dim3(K), dim3(256), 0, stream, stats, K, d, a.D, a.shift, mixing, a.work, a.records, PS,
                           a.info);
    } else {
        const int PS = estep_param_stride(a.D);
        hipLaunchKernelGGL(close_big_finish_kernel<0>, dim3(K, 1 + (a.D + 63) / 64), dim3(256), 0, stream, stats, K, d, a.D, a.shift, mixing, a.work,
                           a.records, PS, a.info);
    }
}

}  // namespace

bool em_close_big_supported(int d) { return d > kMidDim && d <= 1024; }
/// The components' matrices, then room for one parameter set (mixing, means, covariances) and an info block (launch_em_records_big).
size_t em_close_big_work_doubles(int d, int K) { return (size_t)K * big_stride(d) + (size_t)K * ((size_t)d * d + d + 1) + 2 * (size_t)K + 8; }
double* em_close_big_param_area(double* work, int d, int K) { return work + (size_t)K * big_stride(d); }

void launch_em_close_big(const CloseArgs& a, hipStream_t stream)
{
    const int d = a.d, K = a.K;
    hipLaunchKernelGGL(close_big_prepare_kernel, dim3(K, (unsigned)(((size_t)d * d + 255) / 256)), dim3(256), 0, stream, a.stats, K, d, a.shift,
                       a.n_global, a.refine_limit, a.mixing, a.means, a.covs, a.work);
    factor_and_finish(a.stats, a.mixing, a, stream);
}

/// Records of GIVEN parameters: a.mixing / a.means / a.covs are device INPUTS here; a.info receives flags (none) and reach.
void launch_em_records_big(const CloseArgs& a, hipStream_t stream)
{
    const int d = a.d, K = a.K;
    hipLaunchKernelGGL(close_big_from_params_kernel, dim3(K, (unsigned)(((size_t)d * d + 255) / 256)), dim3(256), 0, stream, a.means, a.covs, d,
                       a.work);
    factor_and_finish(nullptr, a.mixing, a, stream);
}

}  // namespace mlhip
