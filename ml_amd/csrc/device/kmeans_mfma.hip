// K-means assignment on the gfx950 fp64 matrix cores with bit-exact labels -- replaces KMeans::assignment_step /
// assign_label (reference ML/KMeans.cpp:153-178) and the sums of update_step (:180-192) for d = 4, 8, ..., 32.
//
// The reference's decision for a sample is  argmin_k fl(sum_j (x_j - c_kj)^2)  with strict '<' (first minimum wins).
// That direct form has no matrix structure, but ranking the clusters does not need it:
//     |x - c_k|^2 = |x|^2 - 2 s_k,      s_k = x . c_k - |c_k|^2 / 2        (larger score = nearer)
// and S = C X is a GEMM. So, per 64-sample group of a wave:
//   1. scores by v_mfma_f64_16x16x4_f64: A = 16 centroids x 4 dims (LDS), B = 4 dims x 16 samples (coordinates held in
//      VGPRs in operand layout), accumulator initialised with -|c|^2/2; every lane tracks best and second-best over the
//      4 x (K/16) clusters it sees. For large K (QUAD) the tracking runs on 32-bit integer keys -- the high dword of the
//      biased, hence positive, score -- and the four scores of an accumulator enter it through their maximum
//      (track_quad_keys): 1.5 integer operations per score instead of 3 fp64 ones plus a compare and a select. The four
//      lane groups are merged by shuffles;
//   2. one lane per sample re-reads the sample's row and evaluates the EXACT direct-form distance to the winner with the
//      same fma chain as the host point query (this is the min distance / inertia contribution that is stored); QUAD: the
//      exact distances to all four clusters of the winning quad, which step 1 did not tell apart;
//   3. if best - second (or best - a mate's score) is not larger than E = (8 (d+2) + 1024) 2^-53 (|x|^2 + max_k |c_k|^2)
//      -- a bound on the rounding error of the two scores plus that of the direct form itself, with a factor 2 to spare,
//      plus the 2 x 2^-44 the tags perturb the two scores by -- the sample is AMBIGUOUS and the lane falls back to the
//      full exact scan (strict '<', ascending k). Otherwise every other cluster is farther than the winner by more than
//      all rounding involved, so the reference's comparison chain picks the same label.
// The labels are therefore bit-exact and the stored distances bit-identical to the VALU kernel (kmeans.hip); only the
// work of the N x K search moves to the matrix pipe (2d flop per pair instead of 3d, no per-cluster compare chain).
//
// Update statistics: exact fixed-point limb sums with integer atomics, exactly as in kmeans.hip.
#include "device.hpp"
#include "parts.hpp"
#include "exact_sum.hpp"

namespace mlhip {
void launch_kmeans_update(const KmeansArgs& a, int grid, size_t pstride, hipStream_t stream);   // kmeans.hip
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BSM = 512;   // threads per workgroup (8 waves; two workgroups per CU)
/// Score of the rows that pad K to a multiple of 16: finite (a tag inserted into -inf's mantissa would make it a NaN),
/// below every real score (|score| <= |x|^2 + max|c|^2 overflows long before it gets here).
constexpr double kPadScore = -0x1p1020;
constexpr int kQuadFactor = 16;   // quad tracking for K >= kQuadFactor * D (and D <= kRegDim)

/// second = max(second, min(best, v)); best = max(best, v) as three bare v_min_f64 / v_max_f64. Written as machine
/// instructions because fmin / fmax on values the compiler cannot prove to be non-signalling (matrix-core results,
/// loop-carried registers) cost an extra canonicalising v_max_f64 each -- 1.5 extra VALU operations per score in a loop
/// whose VALU work is on the critical path. `order` (the argbest select, computed from v by a compiler-visible compare)
/// is only an ordering dependency: the compare is the first reader of the matrix-core result, so the wait states the
/// hardware needs between v_mfma and a VALU read have been inserted by the compiler before this point.
__device__ __forceinline__ void track_top2(double& best, double& second, double v, int order)
{
    double t;
    asm("v_min_f64 %2, %0, %3\n\tv_max_f64 %1, %1, %2\n\tv_max_f64 %0, %0, %3"
        : "+v"(best), "+v"(second), "=&v"(t)
        : "v"(v), "v"(order));
}

/// QUAD tracking on 32-bit integer keys (round 3; before: tagged fp64 scores, v_min / v_max_f64): the scores are biased by
/// max|c|^2 / 2 in the accumulator initialiser, so the score of every cluster a sample can belong to is positive, and the HIGH
/// dword of a positive double orders like the double itself as a signed integer (negative scores -- clusters far behind, or the
/// padding rows -- are negative integers: never ahead of a positive one). The four scores one accumulator holds for a sample
/// (rows r = 0..3: clusters 16 b + g + 4 r) enter the top-2 tracking through the maximum of their keys, and only the BLOCK of
/// the running best is remembered: 6 integer operations per four scores (v_max3_i32, v_max_i32, v_med3_i32, v_cmp_gt_i32,
/// v_cndmask_b32, v_max_i32) instead of 4 byte permutes + 6 fp64 min / max -- next to the matrix
/// instructions a vector instruction costs its issue slot whatever its width (tools/microbench_issue), and these are 30 %
/// fewer. The key keeps 20 mantissa bits: the exact phase turns (best, runner-up) back into a lower / upper bound of the two
/// scores and demands their distance to exceed the rounding margin; the winner's quad (4 clusters) is then settled by exact
/// direct-form distances, everything else falls back to the full scan.
[[maybe_unused]] __device__ __forceinline__ void track_quad_keys(int& best, int& second, int& block, d4 acc, int this_block)
{
    const int k0 = __double2hiint(acc[0]), k1 = __double2hiint(acc[1]), k2 = __double2hiint(acc[2]), k3 = __double2hiint(acc[3]);
    const int m = max(max(k0, k1), max(k2, k3));
    asm("v_med3_i32 %0, %1, %2, %0" : "+v"(second) : "v"(best), "v"(m));   // second <= best: max(second, min(best, m)) is the median
    block = m > best ? this_block : block;
    best = max(best, m);
}

/// One sample's coordinates for the exact phase: in registers (INREGS) or re-read from memory (L1/L2) at every use.
template <int D, bool INREGS> struct SampleRow {
    static constexpr bool kInRegs = INREGS;
    double x[kInRegs ? D : 1];
    const double* p;
    size_t ld;
    __device__ __forceinline__ SampleRow(const double* xt, size_t ldx, uint32_t i) : p(xt + i), ld(ldx)
    {
        if constexpr (kInRegs) {
#pragma unroll
            for (int j = 0; j < D; ++j) x[j] = p[(size_t)j * ld];
        }
    }
    __device__ __forceinline__ double operator()(int j) const
    {
        if constexpr (kInRegs) return x[j];
        else return p[(size_t)j * ld];
    }
};

/// CHUNKED = false: the whole centroid table lives in LDS, waves run independently. CHUNKED = true (tables beyond the LDS
/// budget, i.e. large K): the table is streamed through LDS in chunks of KC clusters; the 8 waves of a workgroup then
/// walk their 64-sample groups in lockstep (two barriers per chunk), keep the running best / second / argbest in
/// registers across chunks, and the exact recheck reads the winner's centroid from global memory (L2).
///
/// QUAD selects how the scores enter the top-2 tracking: tagged quad maxima + mate scores in the exact phase (the fp64
/// VALU work of the scoring loop halves, the exact phase gains 3 d fma per sample: pays when K is large against d), or
/// every score on its own with a separate argbest register (no extra work per sample: small K).
template <int D, bool USE_LDS, bool CHUNKED, bool QUAD>
__global__ __launch_bounds__(BSM, (D <= 16) ? 4 : 2) void kmeans_mfma_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, uint32_t n_pad, int d, const double* __restrict__ cent, int K,
    const double* __restrict__ scale, uint32_t* __restrict__ labels, const uint32_t* __restrict__ old_labels,
    int have_old, double* __restrict__ min_dist, int accumulate, double* __restrict__ partials, size_t pstride, int KC,
    const double* __restrict__ cnorm)
{
    constexpr int Q = D / 4;          // 4-dimension steps of the MFMA
    constexpr int NSB = D <= kMidDim ? 4 : 2;   // 16-sample blocks per wave: the coordinates take Q * NSB doubles per lane
    constexpr int GS = 16 * NSB;      // samples per wave group
    // exact phase: fully unrolled on register-resident coordinates up to d = 64 (d = 32 with the mate scores: five fma
    // chains run over the row, beyond that it spills); above, rolled loops that re-read the sample (unrolling would let
    // the compiler hoist all D loads back into registers)
    constexpr bool kRowInRegs = D <= (QUAD ? kRegDim : kMidDim);
    constexpr int kExactUnroll = kRowInRegs ? D : 8;   // 8 loads in flight per trip: a trip per load is latency-bound. (Late round 5: 16 per trip -- d = 128, K = 64 2.23 -> 2.12 ms but d = 96, K = 256 5.38 -> 6.51, d = 72, K = 1024 13.85 -> 14.37; with 32 the compiler issues the loads ONE at a time, each behind its own s_waitcnt vmcnt(0): 2.23 -> 4.65 ms. 8 kept.)
    constexpr int DS = D + 1;         // odd row stride of the centroid table: conflict-free A-operand reads
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int Kp = (K + 15) & ~15;
    const int KT = CHUNKED ? KC : Kp;          // rows of the LDS table (a multiple of 16)
    double* Cs = smem;                         // [KT][DS] centroids (padding rows zero)
    double* cn = Cs + (size_t)KT * DS;         // [KT]  -|c|^2/2, kPadScore for padding rows
    double* cmax_slot = cn + KT;               // [1]   max_k |c_k|^2
    u64* acc_lds = reinterpret_cast<u64*>(cmax_slot + 1);   // [K][3d+1] when USE_LDS
    __shared__ double red[2 * (BSM / 64)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, s = lane & 15;
    const int W = 3 * d + 1;
    double* my_part = partials + (size_t)blockIdx.x * pstride;
    u64* my_words = reinterpret_cast<u64*>(my_part + 2);

    // ---- workgroup prologue: centroid table, -|c|^2/2 (same ascending-j fma chain as everywhere), max |c|^2, accumulators
    if (accumulate) {
        if (USE_LDS) { for (int e = tid; e < K * W; e += BSM) acc_lds[e] = 0; }
        else         { for (int e = tid; e < K * W; e += BSM) my_words[e] = 0; }
    }
    if constexpr (!CHUNKED) {
        for (int e = tid; e < Kp * DS; e += BSM) {
            const int k = e / DS, j = e - k * DS;
            Cs[e] = (k < K && j < D) ? cent[(size_t)k * D + j] : 0.0;
        }
        __syncthreads();
        for (int k = tid; k < Kp; k += BSM) {
            double nn = 0.0;
            for (int j = 0; j < D; ++j) nn = __builtin_fma(Cs[k * DS + j], Cs[k * DS + j], nn);
            cn[k] = k < K ? -0.5 * nn : kPadScore;
        }
        __syncthreads();
        if (tid < 64) {
            double mx = 0.0;
            for (int k = tid; k < K; k += 64) mx = fmax(mx, -2.0 * cn[k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
            if (tid == 0) *cmax_slot = mx;
        }
    } else {
        double mx = 0.0;
        for (int k = tid; k < K; k += BSM) mx = fmax(mx, -2.0 * cnorm[k]);      // cnorm = -|c|^2/2 (kmeans_cnorm_kernel)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < BSM / 64; ++w) t = fmax(t, red[w]);
            *cmax_slot = t;
        }
    }
    __syncthreads();
    const double cmax2 = *cmax_slot;
    // QUAD: scores biased by max|c|^2 / 2 (>= 0 for every real row; see track_quad_keys)
    if constexpr (QUAD && !CHUNKED) {
        for (int k = tid; k < K; k += BSM) cn[k] += 0.5 * cmax2;
        __syncthreads();
    }
    // rounding of the two scores and of the direct form (8 (d+2) 2^-53, factor 2 to spare) + the tag perturbation of the two
    // scores of the gap (2 x 2^-44 = 1024 x 2^-53)
    const double err_unit = 8.0 * (D + 2) * 0x1p-53;

    double inertia = 0.0, changed = 0.0;
    const uint32_t n_groups = n_pad / GS;
    const uint32_t per_sweep = gridDim.x * (BSM / 64);
    const uint32_t n_sweeps = (n_groups + per_sweep - 1) / per_sweep;          // uniform over the workgroup
    for (uint32_t sweep = 0; sweep < n_sweeps; ++sweep) {
        uint32_t grp = sweep * per_sweep + blockIdx.x * (BSM / 64) + wave;
        const bool active = grp < n_groups;
        if (!active) {
            if constexpr (!CHUNKED) break;     // independent waves: done
            grp = n_groups - 1;                // lockstep: keep serving the barriers (and the table loads) on a valid group
        }
        const uint32_t base = grp * GS;
        // ---- phase 1: scores on the matrix cores. xb[q][sb] = x[dim 4q + g][sample base + 16 sb + s]
        double xb[Q][NSB];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) xb[q][sb] = xt[(size_t)(4 * q + g) * ldx + base + 16 * sb + s];
        // best / second / idx run over all clusters this lane sees; QUAD: integer keys (kbest / ksecond) and the 16-cluster
        // block of the running best in idx
        double best[NSB], second[NSB];
        int kbest[NSB], ksecond[NSB], idx[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            best[sb] = -__builtin_inf();
            second[sb] = -__builtin_inf();
            kbest[sb] = ksecond[sb] = (int)0x80000000;
            idx[sb] = 0;
        }

        __builtin_amdgcn_s_setprio(kMatrixPhasePriority);   // scoring loop over exact phase of the SIMD's other waves (1.25 -> 1.20 ms at d = 8, K = 256)
        for (int k0 = 0; k0 < Kp; k0 += KT) {
        const int rows = min(KT, Kp - k0);      // a multiple of 16
        if constexpr (CHUNKED) {
            __syncthreads();                    // every wave is done with the previous chunk
            // the chunk is one contiguous range of the centroid array: LU independent loads per thread in flight, then the
            // LDS writes with the row padding (column D of a row is never read). LU = 12 covers a 60 KB chunk in ONE round trip
            // (4 until late round 5: three dependent round trips of ~1.5 us per chunk, longer than the chunk's 6 144 cycles of
            // matrix instructions at d = 128 -- the accumulators are dead here, so the registers are there)
            constexpr int LU = 12;
            const int total = rows * D;
            const double* __restrict__ src = cent + (size_t)k0 * D;
            for (int g0 = tid; g0 < total; g0 += LU * BSM) {
                double v[LU];
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int gi = g0 + u * BSM;
                    v[u] = (gi < total && k0 + gi / D < K) ? src[gi] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int gi = g0 + u * BSM;
                    if (gi < total) Cs[(gi / D) * DS + gi % D] = v[u];
                }
            }
            for (int k = tid; k < rows; k += BSM)                            // kPadScore for the padding rows
                cn[k] = (QUAD && k0 + k < K) ? cnorm[k0 + k] + 0.5 * cmax2 : cnorm[k0 + k];
            __syncthreads();
        }
        {
        const int cb0 = 0, cb_end = rows / 16;
        // the four scores (accumulator rows) of block cb for one sample block
        auto consume = [&](int sb, const d4& sc, int cb) {
            double& bst = best[sb];
            double& sec = second[sb];
            int& ix = idx[sb];
            if constexpr (QUAD) {
                track_quad_keys(kbest[sb], ksecond[sb], ix, sc, k0 / 16 + cb);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = sc[r];
                    ix = (v > bst) ? k0 + 16 * cb + g + 4 * r : ix;
                    track_top2(bst, sec, v, ix);
                }
            }
        };
        for (int cb = cb0; cb < cb_end; ++cb) {
            d4 init;
#pragma unroll
            for (int r = 0; r < 4; ++r) init[r] = cn[16 * cb + g + 4 * r];           // D row = (lane>>4) + 4 r
            d4 acc[NSB];
            if constexpr (D <= kMidDim) {
                double a[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) a[q] = Cs[(16 * cb + s) * DS + 4 * q + g];   // A[i = lane&15][k = lane>>4]
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    acc[sb] = init;
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], xb[q][sb], acc[sb], 0, 0, 0);
                }
            } else {
                // large d: the A operands are read step by step (Q of them would not fit next to the coordinates), and
                // TWO cluster blocks advance together where the sub-chunk has a second one: with 2 sample blocks per wave
                // that gives 4 independent accumulator chains instead of 2 (a dependent MFMA waits out the pipeline).
                const bool pair = cb + 1 < cb_end;                  // wave-uniform
                d4 acc2[NSB];
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) acc[sb] = init;
                if (pair) {
                    d4 init2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) init2[r] = cn[16 * (cb + 1) + g + 4 * r];
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb) acc2[sb] = init2;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const double aq = Cs[(16 * cb + s) * DS + 4 * q + g];
                        const double aq2 = Cs[(16 * (cb + 1) + s) * DS + 4 * q + g];
#pragma unroll
                        for (int sb = 0; sb < NSB; ++sb) {
                            acc[sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, xb[q][sb], acc[sb], 0, 0, 0);
                            acc2[sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq2, xb[q][sb], acc2[sb], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int sb = 0; sb < NSB; ++sb) {
                        consume(sb, acc[sb], cb);
                        acc[sb] = acc2[sb];                          // the second block goes through the common tail
                    }
                    ++cb;
                } else {
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const double aq = Cs[(16 * cb + s) * DS + 4 * q + g];
#pragma unroll
                        for (int sb = 0; sb < NSB; ++sb) acc[sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, xb[q][sb], acc[sb], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) consume(sb, acc[sb], cb);
        }
        (void)cb0;
        }
        }   // chunks
        __builtin_amdgcn_s_setprio(0);
        // merge the 4 lane groups (disjoint cluster subsets) of every sample block; lane (g, s) keeps sample 16 g + s
        double my_best = 0.0, my_second = 0.0;
        int my_kbest = 0, my_ksecond = 0;
        int my_idx = 0;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            if constexpr (QUAD) {
                int b = kbest[sb], sd = ksecond[sb];
                int ix = 16 * idx[sb] + g;              // first cluster of the winning quad: 16 block + g (+ 4 r, r = 0..3)
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const int b2 = __shfl_xor(b, off, 64), s2 = __shfl_xor(sd, off, 64), i2 = __shfl_xor(ix, off, 64);
                    const bool take2 = (b2 > b) || (b2 == b && i2 < ix);   // (equal keys: the runner-up equals the best -> ambiguous)
                    sd = max(min(b, b2), max(sd, s2));
                    ix = take2 ? i2 : ix;
                    b = max(b, b2);
                }
                if (g == sb) { my_kbest = b; my_ksecond = sd; my_idx = ix; }
            } else {
                double b = best[sb], sd = second[sb];
                int ix = idx[sb];
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const double b2 = __shfl_xor(b, off, 64), s2 = __shfl_xor(sd, off, 64);
                    const int i2 = __shfl_xor(ix, off, 64);
                    // keep the smaller index on exactly equal scores so that all four lanes agree
                    const bool take2 = (b2 > b) || (b2 == b && i2 < ix);
                    sd = fmax(fmin(b, b2), fmax(sd, s2));
                    ix = take2 ? i2 : ix;
                    b = fmax(b, b2);
                }
                if (g == sb) { my_best = b; my_second = sd; my_idx = ix; }
            }
        }

        // ---- phase 2: one lane per sample, exact arithmetic
        const uint32_t i = base + lane;
        if (active && lane < GS && i < n) {
            const SampleRow<D, kRowInRegs> x(xt, ldx, i);
            double xn = 0.0;
            // (scores that were all NaN never replaced the initial -inf: decoded tag 0 -> cluster g < 16 <= Kp; the clamp keeps
            //  the winner's row inside the table whatever the scores were)
            uint32_t arg = min((uint32_t)my_idx, (uint32_t)(K - 1));
            double dist = 0.0;
            bool certain;
            if constexpr (QUAD) {
                // the four clusters of the winning quad (same block, same lane group: my_idx + 4 r) were told apart by nobody:
                // exact direct-form distances to all of them, the reference's comparison among them (strict '<', ascending k)
                const double* qc[4];
                double qd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t row = min((uint32_t)my_idx + 4u * r, (uint32_t)(K - 1));   // padding rows: any valid row, result unused
                    qc[r] = CHUNKED ? cent + (size_t)row * D : Cs + (size_t)row * DS;
                }
#pragma unroll kExactUnroll
                for (int j = 0; j < D; ++j) {
                    const double v = x(j);
                    xn = __builtin_fma(v, v, xn);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = v - qc[r][j];
                        qd[r] = __builtin_fma(t, t, qd[r]);
                    }
                }
                arg = (uint32_t)my_idx;
                dist = qd[0];
#pragma unroll
                for (int r = 1; r < 4; ++r)
                    if ((uint32_t)my_idx + 4u * r < (uint32_t)K && qd[r] < dist) { dist = qd[r]; arg = (uint32_t)my_idx + 4u * r; }
                // every cluster outside the quad scores below `upper` (its key, one unit up), the quad's best at least `lower`:
                // the quad holds the reference's winner if they are further apart than all rounding involved. A non-positive
                // or non-finite best key (a sample far from every centroid; NaN scores) is never certain.
                const double lower = __hiloint2double(my_kbest, 0);
                const double upper = my_ksecond < 0 ? 0.0 : __hiloint2double(my_ksecond + 1, 0);
                certain = (uint32_t)my_idx < (uint32_t)K && my_kbest > 0 && my_kbest < 0x7ff00000 &&
                          (lower - upper) > err_unit * (xn + cmax2);
            } else {
                // |x|^2 and the exact distance to the winner in one pass over the sample
                const double* c = CHUNKED ? cent + (size_t)arg * D : Cs + (size_t)arg * DS;
#pragma unroll kExactUnroll
                for (int j = 0; j < D; ++j) {
                    const double v = x(j);
                    xn = __builtin_fma(v, v, xn);
                    const double t = v - c[j];
                    dist = __builtin_fma(t, t, dist);
                }
                certain = (my_best - my_second) > err_unit * (xn + cmax2);          // false for NaN / inf as well
            }
            if (!certain) {
                // ambiguous: the reference's own loop (ML/KMeans.cpp:155-163)
                double bd = __builtin_inf();
                uint32_t ba = 0;
                for (int k = 0; k < K; ++k) {
                    const double* c = CHUNKED ? cent + (size_t)k * D : Cs + (size_t)k * DS;
                    double sdist = 0.0;
#pragma unroll kExactUnroll
                    for (int j = 0; j < D; ++j) {
                        const double t = x(j) - c[j];
                        sdist = __builtin_fma(t, t, sdist);
                    }
                    if (sdist < bd) { bd = sdist; ba = (uint32_t)k; }
                }
                arg = ba;
                dist = bd;
            }
            labels[i] = arg;
            if (min_dist) min_dist[i] = dist;
            inertia += dist;
            changed += (!have_old || old_labels[i] != arg) ? 1.0 : 0.0;
            if (accumulate) {
                u64* row = (USE_LDS ? acc_lds : my_words) + (size_t)arg * W;
#pragma unroll kExactUnroll
                for (int j = 0; j < D; ++j) {
                    if (j < d) {
                        u64 w0, w1, w2;
                        split_limbs(x(j) * scale[j], w0, w1, w2);   // wave-uniform index: scalar load
                        atomicAdd(row + 3 * j, w0);
                        atomicAdd(row + 3 * j + 1, w1);
                        atomicAdd(row + 3 * j + 2, w2);
                    }
                }
                atomicAdd(row + 3 * d, (u64)1);
            }
        }
    }
    // block sums of inertia / changed (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        inertia += __shfl_down(inertia, off, 64);
        changed += __shfl_down(changed, off, 64);
    }
    constexpr int NWV = BSM / 64;
    if (lane == 0) {
        red[wave] = inertia;
        red[NWV + wave] = changed;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < NWV; ++w) { a += red[w]; b += red[NWV + w]; }
        my_part[0] = a;
        my_part[1] = b;
    }
    if (accumulate && USE_LDS) {
        for (int e = tid; e < K * W; e += BSM) my_words[e] = acc_lds[e];
    }
}

/// cnorm[k] = -|c_k|^2 / 2 with the ascending-j fma chain used everywhere, kPadScore for the rows that pad K up to a multiple
/// of 16: computed once per step for the chunked-table kernel (per chunk and sweep it would be a serial chain of D
/// dependent loads in front of every barrier).
__global__ __launch_bounds__(256) void kmeans_cnorm_kernel(const double* __restrict__ cent, int K, int Kp, int D,
                                                            double* __restrict__ cnorm)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Kp) return;
    double nn = 0.0;
    if (k < K)
        for (int j = 0; j < D; ++j) nn = __builtin_fma(cent[(size_t)k * D + j], cent[(size_t)k * D + j], nn);
    cnorm[k] = k < K ? -0.5 * nn : kPadScore;
}

template <int D>
int launch_t(const KmeansArgs& a, int num_cus, size_t pstride, hipStream_t stream)
{
    const int Kp = (a.K + 15) & ~15;
    const size_t table = sizeof(double) * ((size_t)Kp * (D + 1) + Kp + 1);
    const size_t accb = sizeof(u64) * (size_t)a.K * (3 * a.d + 1);
    const bool chunked = table > 72 * 1024;                 // table beyond the LDS budget of two workgroups per CU
    const bool use_lds = !chunked && table + accb <= 78 * 1024;
    // Accumulators that do not fit next to the centroid table: assignment only here, the sums by a separate sweep.
    const int accumulate_here = use_lds ? a.accumulate : 0;
    const uint32_t n_pad = padded_samples(a.n);
    int grid = num_cus * 2;
    const uint32_t groups = n_pad / (D <= kMidDim ? 64 : 32);
    const uint32_t need = (groups + BSM / 64 - 1) / (BSM / 64);
    if ((uint32_t)grid > need) grid = (int)(need ? need : 1);
    if ((size_t)grid * pstride > a.partials_capacity) return -2;
    // quad tracking pays when the scoring loop's saving (1.5 K fp64 VALU operations per 64 samples) clearly exceeds the
    // mate scores (3 d fma + 3 d table reads per sample). Measured (kernel time quad / plain, 5M samples): d = 4: 0.96 at
    // K = 64, 0.80 at K = 256; d = 8: 1.15 at K = 32, 1.00 at 64, 0.88 at 256; d = 16: 1.04 at 128, 0.96 at 256; d = 32:
    // 1.03 at 128, 0.98 at 256, 1.00 at 1024; d >= 64: 1.06 .. 1.22 everywhere (the row no longer fits in registers).
    constexpr bool kQuadBuilt = D <= kRegDim;
    const bool quad = kQuadBuilt && a.K >= kQuadFactor * D;
#define MLHIP_KM_ARGS a.xt, a.ldx, a.n, n_pad, a.d, a.centroids, a.K, a.scale, a.labels, a.old_labels, a.have_old, a.min_dist, \
                      accumulate_here, a.partials, pstride
    if (chunked) {
        int KC = (int)(((D <= kMidDim ? 36 : 60) * 1024) / (sizeof(double) * (D + 2))) & ~15;   // rows per chunk: ~36 / 60 KB of table
        if (KC < 16) KC = 16;
        const size_t smem = sizeof(double) * ((size_t)KC * (D + 1) + KC + 1);
        hipLaunchKernelGGL(kmeans_cnorm_kernel, dim3((Kp + 255) / 256), dim3(256), 0, stream, a.centroids, a.K, Kp, D, a.cnorm);
        if (quad) hipLaunchKernelGGL((kmeans_mfma_kernel<D, false, true, kQuadBuilt>), dim3(grid), dim3(BSM), smem, stream, MLHIP_KM_ARGS, KC, a.cnorm);
        else      hipLaunchKernelGGL((kmeans_mfma_kernel<D, false, true, false>), dim3(grid), dim3(BSM), smem, stream, MLHIP_KM_ARGS, KC, a.cnorm);
    } else if (use_lds) {
        if (quad) hipLaunchKernelGGL((kmeans_mfma_kernel<D, true, false, kQuadBuilt>), dim3(grid), dim3(BSM), table + accb, stream, MLHIP_KM_ARGS, 0, a.cnorm);
        else      hipLaunchKernelGGL((kmeans_mfma_kernel<D, true, false, false>), dim3(grid), dim3(BSM), table + accb, stream, MLHIP_KM_ARGS, 0, a.cnorm);
    } else {
        if (quad) hipLaunchKernelGGL((kmeans_mfma_kernel<D, false, false, kQuadBuilt>), dim3(grid), dim3(BSM), table, stream, MLHIP_KM_ARGS, 0, a.cnorm);
        else      hipLaunchKernelGGL((kmeans_mfma_kernel<D, false, false, false>), dim3(grid), dim3(BSM), table, stream, MLHIP_KM_ARGS, 0, a.cnorm);
    }
#undef MLHIP_KM_ARGS
    if (a.accumulate && !use_lds) launch_kmeans_update(a, grid, pstride, stream);
    return grid;
}

}  // namespace

// ---- compiled in six parts by dimension (parts.hpp): 1: D = 4, 8; 2: 12, 16; 3: 20 .. 32; 4: 40 .. 64; 5: 72 .. 96; 6: 104 .. 128
int MLHIP_PART_FN(launch_kmeans_mfma)(const KmeansArgs& a, int num_cus, size_t pstride, hipStream_t stream)
{
    switch (a.D) {
#if MLHIP_PART == 1
    case 4: return launch_t<4>(a, num_cus, pstride, stream);
    case 8: return launch_t<8>(a, num_cus, pstride, stream);
#elif MLHIP_PART == 2
    case 12: return launch_t<12>(a, num_cus, pstride, stream);
    case 16: return launch_t<16>(a, num_cus, pstride, stream);
#elif MLHIP_PART == 3
    case 20: return launch_t<20>(a, num_cus, pstride, stream);
    case 24: return launch_t<24>(a, num_cus, pstride, stream);
    case 28: return launch_t<28>(a, num_cus, pstride, stream);
    case 32: return launch_t<32>(a, num_cus, pstride, stream);
#elif MLHIP_PART == 4
    case 40: return launch_t<40>(a, num_cus, pstride, stream);
    case 48: return launch_t<48>(a, num_cus, pstride, stream);
    case 56: return launch_t<56>(a, num_cus, pstride, stream);
    case 64: return launch_t<64>(a, num_cus, pstride, stream);
#elif MLHIP_PART == 5
    case 72: return launch_t<72>(a, num_cus, pstride, stream);
    case 80: return launch_t<80>(a, num_cus, pstride, stream);
    case 88: return launch_t<88>(a, num_cus, pstride, stream);
    case 96: return launch_t<96>(a, num_cus, pstride, stream);
#elif MLHIP_PART == 6
    case 104: return launch_t<104>(a, num_cus, pstride, stream);
    case 112: return launch_t<112>(a, num_cus, pstride, stream);
    case 120: return launch_t<120>(a, num_cus, pstride, stream);
    case 128: return launch_t<128>(a, num_cus, pstride, stream);
#endif
    default: return -1;
    }
}

#if MLHIP_PART == 1
int launch_kmeans_mfma_part2(const KmeansArgs&, int, size_t, hipStream_t);
int launch_kmeans_mfma_part3(const KmeansArgs&, int, size_t, hipStream_t);
int launch_kmeans_mfma_part4(const KmeansArgs&, int, size_t, hipStream_t);
int launch_kmeans_mfma_part5(const KmeansArgs&, int, size_t, hipStream_t);
int launch_kmeans_mfma_part6(const KmeansArgs&, int, size_t, hipStream_t);

/// The matrix-core kernel handles D = 4, 8, ..., 32, 40, ..., 128 and any K (tables beyond the LDS budget are streamed in chunks).
bool kmeans_mfma_supported(int D, int K)
{
    (void)K;
    return D >= 4 && D <= kMaxDim && D % 4 == 0;
}

int launch_kmeans_mfma(const KmeansArgs& a, int num_cus, hipStream_t stream)
{
    const size_t pstride = 2 + (size_t)a.K * (3 * a.d + 1);
    if (a.D <= 8) return launch_kmeans_mfma_part1(a, num_cus, pstride, stream);
    if (a.D <= 16) return launch_kmeans_mfma_part2(a, num_cus, pstride, stream);
    if (a.D <= 32) return launch_kmeans_mfma_part3(a, num_cus, pstride, stream);
    if (a.D <= 64) return launch_kmeans_mfma_part4(a, num_cus, pstride, stream);
    if (a.D <= 96) return launch_kmeans_mfma_part5(a, num_cus, pstride, stream);
    return launch_kmeans_mfma_part6(a, num_cus, pstride, stream);
}
#endif

}  // namespace mlhip
