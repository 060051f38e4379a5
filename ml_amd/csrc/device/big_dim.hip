// 128 < d <= 1024 on the matrix cores. The tuned kernels of this library keep a sample's coordinates in registers and stop at
// d = 128; the plain tier behind them (generic_dim.hip: one lane per sample, O(d^2) loads per (sample, component), no matrix
// instructions) runs at 0.2 - 0.8 TFLOP/s. The reference has no limit on d (ML/EM.cpp:96-101), and at these dimensions both passes
// of an EM iteration are plain matrix products:
//   * E-step (EM::expectation_step, ML/EM.cpp:190-219): per component Y = W (X - mu) with W = L^-1 lower triangular, q_i = |y_i|^2.
//     One wave per 16 samples: the centred tile Z (d x 16) sits in LDS, W streams through as the A operand of v_mfma_f64_16x16x4
//     row block by row block (16 rows x the columns up to the diagonal), the squares of the 16 x 16 output block are folded into
//     q on the spot -- Y never exists. Same whitening form as every other E-step here: no cancellation.
//   * statistics (EM::maximisation_step, ML/EM.cpp:229-248): S_k = sum_i r_ik x~_i x~_i^T, x~ = [x - shift ; 1], lower triangle.
//     A workgroup owns one 64 x 64 tile of one component's S over one range of samples: panels of 32 samples of the 64 + 64 rows
//     go through LDS (the A panel scaled by r), four waves hold 2 x 2 output blocks each. The sample ranges of a tile are
//     separate partial blocks in the packed [K][F] layout, combined in fixed order by em_reduce_kernel like every other kernel's.
// Records, statistics layout and everything around the two kernels (closing arithmetic, reductions, labels) are the plain tier's.
// MLHIP_BIG_DIM=0: the plain tier instead (A/B runs, tests).
#include <cstdlib>

#include "device.hpp"
#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"

namespace mlhip {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kBigMaxDim = 1024;      // (the E-step's centred tile: 128 KB of the CU's 160 KB of LDS)

/// Lanes l, l ^ 16, l ^ 32, l ^ 48 hold the partial sums of one sample: all four get the total (fixed order).
__device__ __forceinline__ double quad_total(double v)
{
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    const unsigned lo2 = __double2loint(v), hi2 = __double2hiint(v);
    l = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    h = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

/// Records: estep_param_stride(D) doubles per component, [ mean(D) | W packed lower triangle, row by row | coef ] (layout.hpp).
/// A workgroup of four waves per 16 samples: ONE centred tile Z[D][16] in LDS serves all four, which share the row blocks of W
/// (block rb needs rb + 1 column groups: dealt out in a snake, rounds of four, so that every wave gets the same number of
/// products); the four partial sums of |y|^2 meet in LDS in wave order. (One wave per tile keeps a CU at four waves -- the tile
/// is 32 KB at d = 256 -- and every product waits for its own gather of W: 7.7 ms at N = 100k, d = 256, K = 8 against 2.1 here.)
__global__ __launch_bounds__(256) void em_estep_big_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n, uint32_t n_pad, int D,
                                                            const double* __restrict__ params, int K, double* __restrict__ lw_out,
                                                            size_t ldr, double* __restrict__ lse_out, double* __restrict__ ll_partials)
{
    extern __shared__ __attribute__((aligned(16))) double zt[];       // [D][16], then the partial sums qs[4][16]
    double* qs = zt + (size_t)D * 16;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t PS = (size_t)D + (size_t)D * (D + 1) / 2 + 1;
    const int RB = (D + 15) / 16;
    double ll_acc = 0.0;
    const uint32_t n_blocks = n_pad / 16;
    for (uint32_t sb = blockIdx.x; sb < n_blocks; sb += gridDim.x) {
        const uint32_t i0 = sb * 16;
        double m = -__builtin_inf(), s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double* __restrict__ p = params + (size_t)k * PS;
            const double* __restrict__ w = p + D;
            __syncthreads();                                           // the previous component's reads of the tile and of qs are done
            for (int l = tid >> 4; l < D; l += 16) zt[l * 16 + j] = xt[(size_t)l * ldx + i0 + j] - p[l];
            __syncthreads();
            double q = 0.0;
            for (int round = 0; round * 4 < RB; ++round) {
                const int rb = round * 4 + ((round & 1) ? 3 - wave : wave);
                if (rb >= RB) continue;                                // (wave-uniform)
                const int row = rb * 16 + j;                           // (the A operand's row index is the lane's low four bits too)
                const bool row_ok = row < D;
                const double* __restrict__ wr = w + (size_t)row * (row + 1) / 2;
                const int l_end = (rb * 16 + 16 < D ? rb * 16 + 16 : D);   // columns 0 .. l_end - 1 (a multiple of 4)
                d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                // The columns LEFT of the diagonal block need no triangle test, and only the last row block can hold rows >= D: the
                // bulk of the products runs without predicates, the lane's row pointer and its LDS address advancing by constants
                // (13 vector instructions per product before -- address arithmetic and selects --, counters in profiles/r04_pmc_new_kernels.txt).
                const double* __restrict__ wp = wr + kq;
                const double* zp = zt + kq * 16 + j;
                const int l_full = rb * 16;                            // (a multiple of 16)
                int l0 = 0;
                if (rb * 16 + 16 <= D) {
                    for (; l0 < l_full; l0 += 16) {                     // four gathers of W in flight
                        const double a0 = wp[l0], a1 = wp[l0 + 4], a2 = wp[l0 + 8], a3 = wp[l0 + 12];
                        const double b0 = zp[l0 * 16], b1 = zp[(l0 + 4) * 16], b2 = zp[(l0 + 8) * 16], b3 = zp[(l0 + 12) * 16];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc1, 0, 0, 0);
                    }
                }
                for (; l0 + 8 <= l_end; l0 += 8) {                      // the diagonal block (and all of the last row block)
                    const int c0 = l0 + kq, c1 = l0 + 4 + kq;
                    const double a0 = (row_ok && c0 <= row) ? wr[c0] : 0.0;
                    const double a1 = (row_ok && c1 <= row) ? wr[c1] : 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, zt[c0 * 16 + j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, zt[c1 * 16 + j], acc1, 0, 0, 0);
                }
                for (; l0 < l_end; l0 += 4) {
                    const int c0 = l0 + kq;
                    const double a0 = (row_ok && c0 <= row) ? wr[c0] : 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, zt[c0 * 16 + j], acc0, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double y = acc0[g] + acc1[g];                // rows kq + 4 g of the block, sample j
                    q = __builtin_fma(y, y, q);
                }
            }
            q = quad_total(q);
            if (kq == 0) qs[wave * 16 + j] = q;
            __syncthreads();
            if (wave == 0) {
                const double qt = ((qs[j] + qs[16 + j]) + qs[32 + j]) + qs[48 + j];
                const double lw = __builtin_fma(-0.5, qt, p[PS - 1]);
                if (kq == 0) lw_out[(size_t)k * ldr + i0 + j] = lw;
                const double e = exp_nonpos(lw == -HUGE_VAL ? -HUGE_VAL : -fabs(lw - m));   // online log-sum-exp (em_estep.hip)
                const bool up = lw > m;
                s = up ? __builtin_fma(s, e, 1.0) : s + e;
                m = up ? lw : m;
            }
        }
        if (wave == 0) {
            const double lse = m + log(s);
            if (kq == 0) {
                lse_out[i0 + j] = lse;
                if (i0 + j < n) ll_acc += lse;
            }
        }
    }
    // lanes 0 .. 15 of wave 0 carry the sums of their sample column; fixed-order tree over them
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) ll_acc += __shfl_down(ll_acc, off, 64);
    if (tid == 0) ll_partials[blockIdx.x] = ll_acc;
}

// ---- E-step, tiled as a matrix product (round 5) ------------------------------------------------------------------------
// The kernel above keeps ONE centred tile of 16 samples per workgroup (all D rows of it: 128 KB at D = 1024) and streams the whole of W
// past it: every element of W is fetched again for every 16 samples -- 4 flop per byte from L2, which is what it ran at (15.8 TFLOP/s
// at d = 1024: 52 GB through L2 per launch). Here a workgroup owns a tile of 128 rows of Y = W (X - mu) x 128 samples in registers (each
// of four waves 64 x 64: 16 accumulator blocks -- its rows are every other 16-row block of the tile) and walks the columns l of W in chunks of 16: 128 x 16 of W and 16 x 128 of the centred
// samples go through LDS, double-buffered, the next chunk in flight during the matrix phase: 16 flop per byte. Only chunks on or left
// of the tile's diagonal exist; a wave skips the 16-row blocks that lie wholly right of it. The squares of a finished tile are folded
// into the samples' q on the spot; log-sum-exp over the components is a separate pass over lw (em_lse_rows_kernel), so a unit of work
// is (sample tile, component) and the units are dealt to persistent workgroups round-robin.
constexpr int GR = 128, GS = 128, GC = 16;      // rows, samples, columns per chunk
constexpr int GWS = GC + 1;                     // LDS row stride of the W chunk (odd)
constexpr int GZS = GS + 16;                    // ... of the centred-sample chunk: 16 mod 32 doubles -- the B operand's two rows of a 32-lane
                                                // half (kq = 0, 1) land 32 banks apart; with GS + 1 they were ONE bank pair apart and 15 of 16
                                                // lanes collided (SQ_LDS_BANK_CONFLICT 4.6e8 of SQ_LDS_IDX_ACTIVE 7.4e8 at d = 1024)

///
/// P > 1 (late round 5): with FEW units -- a small N K against the chip's 2 x CUs workgroup slots: 1 564 units are 3.05 rounds run as 4 at
/// N = 50k, d = 1024, K = 4, and a test-sized fit leaves most of the chip idle -- a unit's row blocks are dealt to P parts (part p takes the
/// blocks p, 2P - 1 - p, 2P + p, 4P - 1 - p, ...: the triangle's cost in equal shares when the count is a multiple of 2P), a part being a
/// unit of its own that leaves its share of q in q_parts[part][k][i]; em_lse_rows_kernel adds the shares in ascending part order and
/// forms lw. P = 1 is the form above, bit for bit.
__global__ __launch_bounds__(256, 2) void em_estep_gemm_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n_pad, int D,
                                                                const double* __restrict__ params, int K, double* __restrict__ lw_out,
                                                                size_t ldr, int P, double* __restrict__ q_parts)
{
    __shared__ double Wc[2][GR * GWS];
    __shared__ double Zc[2][GC * GZS];
    __shared__ double qs[2][GS];
    const int tid = threadIdx.x, lane = tid & 63, i_r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;                           // the wave's 64 rows x 64 samples of the tile
    const size_t PS = (size_t)D + (size_t)D * (D + 1) / 2 + 1;
    const int n_rb = (D + GR - 1) / GR;
    const uint32_t n_tiles = n_pad / GS, per_part = n_tiles * (uint32_t)K, n_units = per_part * (uint32_t)P;
    // staging roles: W chunk -- row w_r, columns 8 w_h .. + 7; Z chunk -- column (of W) z_j, samples z_g + 16 j (a thread's eight samples
    // interleaved with its neighbours': 16 lanes load and store 16 consecutive doubles -- eight consecutive ones per thread put the lanes
    // of a store 64 bytes apart, an 8-way bank conflict)
    const int w_r = tid >> 1, w_h = tid & 1, z_j = tid >> 4, z_g = tid & 15;
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int part = (int)(unit / per_part);                       // (part-major: the parts of a tile are spread over the grid)
        const uint32_t tk = unit - (uint32_t)part * per_part;
        const uint32_t tile = tk / (uint32_t)K;
        const int k = (int)(tk - tile * (uint32_t)K);
        const uint32_t i0 = tile * GS;
        // the part's row blocks in ascending order: m-th is m P + part (m even), (m + 1) P - 1 - part (m odd)
        auto rb_at = [&](int m) { return (m & 1) ? (m + 1) * P - 1 - part : m * P + part; };
        const double* __restrict__ p = params + (size_t)k * PS;
        const double* __restrict__ w = p + D;
        double qacc[4] = {0.0, 0.0, 0.0, 0.0};
        double wv[8], zv[8], mu = 0.0;
        // chunk (rb, c): rows rb * 128 .., columns 16 c ..; requested into registers one chunk ahead
        auto prefetch = [&](int rb, int c) {
            size_t ldx_ = ldx;
            asm volatile("" : "+s"(ldx_));                              // (addresses formed per call, not carried through the matrix phase)
            const int row = rb * GR + w_r, l0 = c * GC + 8 * w_h;
            const int rowc = row < D ? row : D - 1;
            const double* __restrict__ wr = w + (size_t)rowc * (rowc + 1) / 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = wr[l0 + j <= rowc ? l0 + j : rowc];
            const int l = c * GC + z_j, lc = l < D ? l : D - 1;
            const double* __restrict__ xr = xt + (size_t)lc * ldx_ + i0 + z_g;
#pragma unroll
            for (int j = 0; j < 8; ++j) zv[j] = xr[16 * j];
            mu = p[lc];
        };
        auto stage = [&](int rb, int c, int buf) {
            const int row = rb * GR + w_r, l0 = c * GC + 8 * w_h;
#pragma unroll
            for (int j = 0; j < 8; ++j) Wc[buf][w_r * GWS + 8 * w_h + j] = (row < D && l0 + j <= row) ? wv[j] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) Zc[buf][z_j * GZS + z_g + 16 * j] = zv[j] - mu;
        };
        int buf = 0;
        prefetch(rb_at(0), 0);                                          // (part < P <= n_rb: the part's first block exists)
        for (int m = 0, rb = rb_at(0); rb < n_rb; ++m, rb = rb_at(m)) {
            const int row0 = rb * GR;
            const int rb_next = rb_at(m + 1);
            const int l_end = row0 + GR < D ? row0 + GR : D;             // columns 0 .. l_end - 1
            const int n_c = (l_end + GC - 1) / GC;
            d4 acc[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < n_c; ++c, buf ^= 1) {
                stage(rb, c, buf);
                __syncthreads();
                if (c + 1 < n_c) prefetch(rb, c + 1);
                else if (rb_next < n_rb) prefetch(rb_next, 0);
                const double* Wb = Wc[buf] + (wa * 16 + i_r) * GWS + kq;
                const double* Zb = Zc[buf] + kq * GZS + wb * 64 + i_r;
                // block u of this wave: rows first_row + 32 u .. + 15 -- the tile's eight 16-row blocks dealt ALTERNATELY to the two
                // waves of a sample half. With 64 consecutive rows per wave the lower wave carried the whole triangle of a diagonal tile
                // and the upper one all of a ragged last tile (d = 192: rows 128 .. 191 on one wave, the other idle), and the waves of
                // a kind sit on the same SIMDs in every workgroup of the CU.
                const int first_row = row0 + wa * 16;
#pragma unroll
                for (int ks = 0; ks < GC / 4; ++ks) {
                    const int l_lo = c * GC + 4 * ks;                   // columns l_lo .. l_lo + 3
                    double bv[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) bv[v] = Zb[4 * ks * GZS + 16 * v];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (l_lo <= first_row + 32 * u + 15 && first_row + 32 * u < D) {   // (wave-uniform: blocks wholly right of the diagonal or below row D - 1 hold zeros)
                            const double av = Wb[32 * u * GWS + 4 * ks];
#pragma unroll
                            for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[v], acc[u][v], 0, 0, 0);
                        }
                    }
                }
            }
            // the finished tile's squares: lane (i_r, kq) holds rows 16 wa + kq + 4 g + 32 u of sample 16 v + i_r
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) qacc[v] = __builtin_fma(acc[u][v][g], acc[u][v][g], qacc[v]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double q = quad_total(qacc[v]);
            if (kq == 0) qs[wa][wb * 64 + 16 * v + i_r] = q;
        }
        __syncthreads();
        if (tid < GS) {
            if (P == 1) lw_out[(size_t)k * ldr + i0 + tid] = __builtin_fma(-0.5, qs[0][tid] + qs[1][tid], p[PS - 1]);
            else q_parts[((size_t)part * K + k) * ldr + i0 + tid] = qs[0][tid] + qs[1][tid];
        }
        __syncthreads();                                               // (qs and the LDS buffers are reused by the next unit)
    }
}

/// lse_i = log sum_k exp(lw_ki) (the online form of em_estep.hip, k ascending) and the per-workgroup sums of lse over the live samples.
/// P > 1: lw is formed here from the parts' shares of q (ascending part order) and the records' constants, and written.
__global__ __launch_bounds__(256) void em_lse_rows_kernel(double* __restrict__ lw, size_t ldr, uint32_t n, uint32_t n_pad, int K,
                                                           double* __restrict__ lse_out, double* __restrict__ ll_partials, int P,
                                                           const double* __restrict__ q_parts, const double* __restrict__ params, size_t PS)
{
    __shared__ double red[4];
    double ll_acc = 0.0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_pad; i += gridDim.x * 256u) {
        double m = -__builtin_inf(), s = 0.0;
        for (int k = 0; k < K; ++k) {
            double v;
            if (P == 1) v = lw[(size_t)k * ldr + i];
            else {
                double q = q_parts[(size_t)k * ldr + i];
                for (int part = 1; part < P; ++part) q += q_parts[((size_t)part * K + k) * ldr + i];
                v = __builtin_fma(-0.5, q, params[(size_t)k * PS + PS - 1]);
                lw[(size_t)k * ldr + i] = v;
            }
            const double e = exp_nonpos(v == -HUGE_VAL ? -HUGE_VAL : -fabs(v - m));
            const bool up = v > m;
            s = up ? __builtin_fma(s, e, 1.0) : s + e;
            m = up ? v : m;
        }
        const double lse = m + log(s);
        lse_out[i] = lse;
        if (i < n) ll_acc += lse;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ll_acc += __shfl_down(ll_acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ll_acc;
    __syncthreads();
    if (threadIdx.x == 0) ll_partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- statistics ----------------------------------------------------------------------------------------------------------
constexpr int MT = 64;        // macro tile: 64 x 64 entries of S_k
constexpr int SC = 32;        // samples per panel
constexpr int XS = 33;        // LDS row stride of a panel: ODD. The compiler pairs the reads of two sample groups into ds_read2_b64, which is
                              // served in 16-lane groups over 32 banks of 4 bytes: the 16 rows of a group must start on 16 different even
                              // banks (round 4's 34 suited ds_read_b64's 32-lane halves over 64 banks and cost 28 % of the LDS cycles in
                              // conflicts: SQ_LDS_BANK_CONFLICT 3.96e7 of SQ_LDS_IDX_ACTIVE 1.41e8 at N = 100k, d = 256, K = 8)

/// grid: (macro tile pair t, group of KC components, sample range s). Tile pair t = (ta, tb), tb <= ta, row-major over the lower
/// triangle. The panels of X are the same for every component -- only the responsibilities differ -- so ONE pair of panels in LDS
/// serves KC components: the A operand is the raw x~ scaled by the component's r on its way into the matrix instruction. (One
/// component per workgroup re-reads the panels K times: 12 GB through L2 per launch at N = 100k, d = 256, K = 8 -- 2.7 TB/s, which is
/// what bound the first form of this kernel at 4.4 ms.)
constexpr int KC = 4;

/// Macro tiles along one edge of S_k: over the d COORDINATES only. The row of the constant 1 (the S1 sums and S0) would open a tile row
/// of its own whenever d is a multiple of 64 -- 15 tile pairs instead of 10 at d = 256, all of it padding but one row; it is formed by
/// the workgroups of the DIAGONAL tiles instead, on the vector unit, from the panel and the responsibilities they hold in LDS anyway:
/// thread (row r of the tile, component c of the group) adds r_ic x~_ir over the panel's 32 samples in order -- 32 fused multiply-adds per
/// chunk next to 8 192 cycles of matrix instructions (round 5; a separate kernel for that row took 0.5 - 0.7 ms: it re-read the
/// responsibilities per 64 columns with no load in flight during its arithmetic).
__host__ __device__ inline int big_tiles(int d) { return (d + MT - 1) / MT; }

__global__ __launch_bounds__(256, 2) void em_mstats_big_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n, int d,
                                                             const double* __restrict__ shift, const double* __restrict__ lw,
                                                             size_t ldr, const double* __restrict__ lse, int mode,
                                                             double* __restrict__ partials, int K, int F, uint32_t chunks_per_split)
{
    __shared__ double pa[MT * XS], pb[MT * XS];
    __shared__ double rr[KC * SC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ta = 0;
    while ((ta + 1) * (ta + 2) / 2 <= (int)blockIdx.x) ++ta;
    const int tb = (int)blockIdx.x - ta * (ta + 1) / 2;
    const int k0 = blockIdx.y * KC;
    const int a_base = ta * MT, b_base = tb * MT;
    const int wa = wave >> 1, wb = wave & 1;                            // the wave's 32 x 32 quarter of the macro tile
    const bool idle = ta == tb && wa < wb;                              // above the diagonal: not needed
    d4 acc[KC][2][2];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) acc[c][u][v] = d4{0.0, 0.0, 0.0, 0.0};
    const uint32_t n_chunks = (n + SC - 1) / SC;
    const uint32_t c_begin = blockIdx.z * chunks_per_split;
    uint32_t c_end = c_begin + chunks_per_split;
    if (c_end > n_chunks) c_end = n_chunks;
    const int i_r = lane & 15, kq = lane >> 4;
    const int l_r = tid & (MT - 1), l_c = tid / MT;                     // ones row (diagonal tiles): this thread's coordinate and component
    double s1 = 0.0, s0 = 0.0;
    // Software pipeline (round 5): the NEXT chunk's panel entries and responsibilities are requested right behind the barrier and are in
    // flight during the matrix phase of this chunk; they go to LDS at the top of the next trip. (Round 4 loaded global -> LDS between
    // the two barriers: every chunk exposed a memory round trip, the matrix pipe was 43 % busy.)
    // Staging roles (late round 5): a thread owns ONE row of each panel (p_r = tid / 4) and eight consecutive samples of it (p_q = tid % 4):
    // the row's address, its shift and what a row at or beyond d holds (the constant 1 / zero) are formed ONCE -- 12 registers carried
    // through the loop -- and a chunk costs the thread 2 x 4 16-byte loads, 16 subtractions and 16 LDS writes. (Before: 16 rows per thread,
    // e = tid + 256 t -> (row e / 32, sample e % 32): 16 row clamps, 16 64-bit address products, 16 shift loads, 32 selects and 16 8-byte
    // loads per chunk -- ~250 vector instructions per thread and chunk beside the 128 matrix instructions of its wave, SQ_INSTS_VALU 2.7
    // per matrix instruction; hoisting THEM out of the loop costs 40 registers and spills.)
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int NP = SC / 4 / 2;                                       // 16-byte loads per thread, panel and chunk
    d2 va[NP], vb[NP];
    double vr = 0.0, vl = 0.0;
    const int p_r = tid >> 2, p_q = tid & 3;
    const int p_s = tid & (SC - 1), r_c = tid / SC;                     // responsibilities (tid < KC * SC): sample column, component of the group
    const int ar = a_base + p_r, br = b_base + p_r;
    const bool la = ar < d, lb = br < d;                                // a row of X (rows beyond d - 1 re-read row d - 1: discarded below)
    const double ca = ar == d ? 1.0 : 0.0, cb = br == d ? 1.0 : 0.0;    // the row of the constant 1, padding rows
    const double sa = shift[la ? ar : d - 1], sb = shift[lb ? br : d - 1];
    const d2* __restrict__ pa_src = reinterpret_cast<const d2*>(xt + (size_t)(la ? ar : d - 1) * ldx + 8 * p_q);
    const d2* __restrict__ pb_src = reinterpret_cast<const d2*>(xt + (size_t)(lb ? br : d - 1) * ldx + 8 * p_q);
    auto prefetch = [&](uint32_t ch) {
        const size_t o = (size_t)ch * (SC / 2);                         // (samples ch * 32 .. + 31 < n_pad: the allocation is padded to the tile)
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            va[t] = pa_src[o + t];
            vb[t] = pb_src[o + t];
        }
        if (tid < KC * SC) {
            const int k = k0 + r_c;
            const uint32_t i = ch * SC + p_s;
            vr = lw[(size_t)(k < K ? k : 0) * ldr + i];
            if (mode != kFromResp) vl = lse[i];
        }
    };
    if (c_begin < c_end) prefetch(c_begin);
    for (uint32_t ch = c_begin; ch < c_end; ++ch) {
        const uint32_t i0 = ch * SC;
        __syncthreads();                                               // the previous panels have been consumed
        if (tid < KC * SC) {                                           // responsibilities of the group's components for the 32 samples
            const int k = k0 + r_c;
            rr[tid] = (k < K && i0 + p_s < n) ? (mode == kFromResp ? vr : exp_nonpos(vr - vl)) : 0.0;
        }
        {
            double* __restrict__ wa_ = pa + p_r * XS + 8 * p_q;
            double* __restrict__ wb_ = pb + p_r * XS + 8 * p_q;
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                wa_[2 * t] = la ? va[t][0] - sa : ca;
                wa_[2 * t + 1] = la ? va[t][1] - sa : ca;
                wb_[2 * t] = lb ? vb[t][0] - sb : cb;
                wb_[2 * t + 1] = lb ? vb[t][1] - sb : cb;
            }
        }
        __syncthreads();
        if (ch + 1 < c_end) prefetch(ch + 1);
        if (ta == tb) {                                                // (workgroup-uniform) the ones row: S1 over this tile's 64 coordinates, S0
            const double* __restrict__ rc = rr + l_c * SC;
            const double* __restrict__ xr = pb + l_r * XS;
#pragma unroll 8
            for (int sidx = 0; sidx < SC; ++sidx) {
                s1 = __builtin_fma(rc[sidx], xr[sidx], s1);
                s0 += rc[sidx];
            }
        }
        if (!idle) {
#pragma unroll
            for (int ks = 0; ks < SC / 4; ++ks) {
                double av[2], bv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) av[u] = pa[(wa * 32 + u * 16 + i_r) * XS + ks * 4 + kq];
#pragma unroll
                for (int v = 0; v < 2; ++v) bv[v] = pb[(wb * 32 + v * 16 + i_r) * XS + ks * 4 + kq];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const double r = rr[c * SC + ks * 4 + kq];           // the A operand's sample: 4 ks + (lane >> 4)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const double a = av[u] * r;
#pragma unroll
                        for (int v = 0; v < 2; ++v) acc[c][u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[v], acc[c][u][v], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (ta == tb && k0 + l_c < K) {
        double* __restrict__ out = partials + ((size_t)blockIdx.z * K + k0 + l_c) * F + (size_t)d * (d + 1) / 2;
        if (b_base + l_r < d) out[b_base + l_r] = s1;
        if (ta == 0 && l_r == 0) out[d] = s0;
    }
    if (idle) return;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = k0 + c;
        if (k >= K) break;
        double* __restrict__ out = partials + ((size_t)blockIdx.z * K + k) * F;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int a = a_base + wa * 32 + u * 16 + kq + 4 * g;   // output row: kq + 4 g of the block; column: lane & 15
                    const int b = b_base + wb * 32 + v * 16 + i_r;
                    if (a < d && b <= a) out[(size_t)a * (a + 1) / 2 + b] = acc[c][u][v][g];
                }
    }
}

// ---- K-means assignment -----------------------------------------------------------------------------------------------
/// KMeans::assignment_step / assign_label (ML/KMeans.cpp:153-178) for d > 128: the reference's own arithmetic -- per
/// (sample, cluster) the ascending-j chain s = fma(x_j - c_kj, x_j - c_kj, s), strict '<' over ascending k -- so labels and
/// distances are bit for bit the plain tier's. What changes is the traffic: the plain kernel re-reads a
/// sample's d coordinates from memory for every cluster; here a lane holds the running sums of SIXTEEN clusters for TWO samples
/// and walks the dimensions once per such block -- one coordinate load per 64 fused multiply-adds -- with the 16 centroid
/// coordinates of a dimension arriving as ONE scalar load from a dimension-major copy of the table ([D][Kp], built per launch).
constexpr int KB = 16;        // clusters per register block

__global__ __launch_bounds__(256) void kmeans_transpose_kernel(const double* __restrict__ cent, int K, int Kp, int D, double* __restrict__ centT)
{
    for (int e = blockIdx.x * 256 + threadIdx.x; e < D * Kp; e += gridDim.x * 256) {
        const int j = e / Kp, k = e - j * Kp;
        centT[e] = k < K ? cent[(size_t)k * D + j] : 0.0;
    }
}

__global__ __launch_bounds__(256) void kmeans_assign_big_kernel(const double* __restrict__ xt, size_t ldx, uint32_t n, uint32_t n_pad, int D,
                                                                 const double* __restrict__ centT, int Kp, int K,
                                                                 uint32_t* __restrict__ labels, const uint32_t* __restrict__ old_labels,
                                                                 int have_old, double* __restrict__ min_dist,
                                                                 double* __restrict__ partials, size_t pstride)
{
    __shared__ double red[8];
    double inertia = 0.0, changed = 0.0;
    for (uint32_t base = blockIdx.x * 512u; base < n_pad; base += gridDim.x * 512u) {
        const uint32_t i0 = base + threadIdx.x, i1 = i0 + 256u;          // samples of this lane (stored only below n)
        const uint32_t l0 = i0 < n_pad ? i0 : n_pad - 1, l1 = i1 < n_pad ? i1 : n_pad - 1;   // ... read inside the allocation
        double best0 = __builtin_inf(), best1 = __builtin_inf();
        uint32_t arg0 = 0, arg1 = 0;
        for (int kc = 0; kc < Kp; kc += KB) {
            double s0[KB], s1[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) s0[u] = s1[u] = 0.0;
            const double* __restrict__ ct = centT + kc;                   // wave-uniform: scalar loads
#pragma unroll 2
            for (int j = 0; j < D; ++j) {
                const double x0 = xt[(size_t)j * ldx + l0], x1 = xt[(size_t)j * ldx + l1];
                const double* __restrict__ cj = ct + (size_t)j * Kp;
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const double c = cj[u];
                    const double t0 = x0 - c, t1 = x1 - c;
                    s0[u] = __builtin_fma(t0, t0, s0[u]);
                    s1[u] = __builtin_fma(t1, t1, s1[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                if (kc + u < K) {                                         // (wave-uniform)
                    if (s0[u] < best0) { best0 = s0[u]; arg0 = (uint32_t)(kc + u); }
                    if (s1[u] < best1) { best1 = s1[u]; arg1 = (uint32_t)(kc + u); }
                }
            }
        }
        if (i0 < n) {
            labels[i0] = arg0;
            if (min_dist) min_dist[i0] = best0;
            inertia += best0;
            changed += (!have_old || old_labels[i0] != arg0) ? 1.0 : 0.0;
        }
        if (i1 < n) {
            labels[i1] = arg1;
            if (min_dist) min_dist[i1] = best1;
            inertia += best1;
            changed += (!have_old || old_labels[i1] != arg1) ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        inertia += __shfl_down(inertia, off, 64);
        changed += __shfl_down(changed, off, 64);
    }
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

bool big_dim_enabled()
{
    const char* e = std::getenv("MLHIP_BIG_DIM");
    return !(e && e[0] == '0');
}

}  // namespace

bool big_dim_applies(int d) { return d > kMaxDim && d <= kBigMaxDim && big_dim_enabled(); }
/// The K-means assignment kernel of this file has no upper limit on d (no tile in LDS).
bool big_dim_kmeans_applies(int d) { return d > kMaxDim && big_dim_enabled(); }

constexpr int kMaxSplits = 64;      // (32 until late in round 5: 12 units x 32 = 384 workgroups on 512 slots at d = 192, K = 8)
/// Sample ranges a statistics tile is cut into (= partial blocks written), at most kMaxSplits: the count whose workgroups fill whole rounds of
/// the chip's 2 x CUs resident slots best (round 4 took ceil(3 CUs / units): 20 units x 32 = 640 workgroups on 512 slots at d = 256,
/// K = 8 -- a second round a quarter full), the smallest such count within 3 % of the best.
int big_dim_splits(int d, int K, int num_cus)
{
    const int T = big_tiles(d);
    const int units = T * (T + 1) / 2 * ((K + KC - 1) / KC);            // (tile pair, component group)
    const int slots = 2 * num_cus;
    double best = 0.0;
    for (int s = 1; s <= kMaxSplits; ++s) {
        const int wgs = units * s, rounds = (wgs + slots - 1) / slots;
        const double eff = (double)wgs / ((double)rounds * slots);
        if (eff > best) best = eff;
    }
    for (int s = 1; s <= kMaxSplits; ++s) {
        const int wgs = units * s, rounds = (wgs + slots - 1) / slots;
        if ((double)wgs / ((double)rounds * slots) >= best - 0.03 && wgs >= slots) return s;
    }
    for (int s = 1; s <= kMaxSplits; ++s) {                                     // (fewer workgroups than slots whatever the count: the fullest)
        const int wgs = units * s, rounds = (wgs + slots - 1) / slots;
        if ((double)wgs / ((double)rounds * slots) >= best - 1e-9) return s;
    }
    return 1;
}

/// Parts a unit of the tiled E-step is cut into (1, 2, 4 or 8; at most the number of row blocks and what the scratch block holds): the
/// count with the shortest schedule when the units are dealt round-robin to `slots` workgroups, a part costing the chunks of its row
/// blocks (block rb: the chunks left of and on its diagonal, its rows inside W) plus a fixed share per unit (prologue, the q write).
int estep_parts(uint32_t units1, int D, int slots, size_t max_parts)
{
    const int n_rb = (D + GR - 1) / GR;
    if (const char* e = ab_env("MLHIP_ESTEP_PARTS")) {                          // (A/B runs: a given count where it is allowed)
        const int want = std::atoi(e);
        if (want >= 1 && want <= 8 && want <= n_rb && (want == 1 || (size_t)want <= max_parts)) return want;
    }
    if (units1 == 0 || units1 >= 16u * (uint32_t)slots) return 1;              // many rounds: the tail is a few per cent at most
    int best_p = 1;
    double best_t = 0.0;
    for (int P = 1; P <= 8 && P <= n_rb && (P == 1 || (size_t)P <= max_parts); P *= 2) {
        double cost[8];
        for (int part = 0; part < P; ++part) {
            double c = 4.0;                                                     // (fixed share, in chunks)
            for (int m = 0;; ++m) {
                const int rb = (m & 1) ? (m + 1) * P - 1 - part : m * P + part;
                if (rb >= n_rb) break;
                const int l_end = rb * GR + GR < D ? rb * GR + GR : D;
                const int rows = l_end - rb * GR;
                c += (double)((l_end + GC - 1) / GC) * (rows > 64 ? 1.0 : 0.6);
            }
            cost[part] = c;
        }
        // round-robin: workgroup w takes the units w, w + slots, ...; unit u belongs to part u / units1
        const uint64_t total = (uint64_t)units1 * P;
        double t = 0.0;
        const uint32_t probe = total < (uint64_t)slots ? (uint32_t)total : (uint32_t)slots;
        for (uint32_t w = 0; w < probe; w += (probe > 64 ? probe / 64 : 1)) {   // (a sample of the workgroups; the first ones carry the longest lists)
            double tw = 0.0;
            for (uint64_t u = w; u < total; u += (uint64_t)slots) tw += cost[u / units1];
            if (tw > t) t = tw;
        }
        if (best_t == 0.0 || t < 0.97 * best_t) { best_t = t; best_p = P; }
    }
    return best_p;
}

int launch_em_estep_big(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t n_pad = padded_samples(a.n);
    static const bool tiled = [] { const char* e = ab_env("MLHIP_ESTEP_BIG"); return !(e && e[0] == 't'); }();   // "tile": round 4's kernel (A/B)
    if (tiled) {
        static_assert(kSampleTile % GS == 0, "a sample tile of the product must divide the padding granule of N");
        const int P = estep_parts(n_pad / GS * (uint32_t)a.K, a.D, 2 * num_cus, a.scratch ? a.scratch_doubles / ((size_t)a.K * a.ldr) : 0);
        const uint32_t units = n_pad / GS * (uint32_t)a.K * (uint32_t)P;
        uint32_t grid = 2u * (uint32_t)num_cus;
        if (grid > units) grid = units;
        hipLaunchKernelGGL(em_estep_gemm_kernel, dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, n_pad, a.D, a.params, a.K, a.lw, a.ldr, P, a.scratch);
        uint32_t lgrid = (n_pad + 255) / 256;
        if (lgrid > (uint32_t)a.n_ll_partials) lgrid = (uint32_t)a.n_ll_partials;
        if (lgrid > 4u * (uint32_t)num_cus) lgrid = 4u * (uint32_t)num_cus;
        hipLaunchKernelGGL(em_lse_rows_kernel, dim3(lgrid), dim3(256), 0, stream, a.lw, a.ldr, a.n, n_pad, a.K, a.lse, a.ll_partials, P,
                           (const double*)a.scratch, a.params, (size_t)a.D + (size_t)a.D * (a.D + 1) / 2 + 1);
        return (int)lgrid;
    }
    const size_t smem = sizeof(double) * ((size_t)a.D * 16 + 64);     // 64.5 KB at D = 512, 128.5 KB at D = 1024
    if (smem > 64 * 1024 &&                                           // (per device: asked for on every such launch)
        hipFuncSetAttribute(reinterpret_cast<const void*>(em_estep_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess)
        return -1;
    const int per_cu = (int)(size_t(160 * 1024) / (smem + 1024));
    uint32_t grid = (uint32_t)num_cus * (uint32_t)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    const uint32_t blocks = n_pad / 16;
    if (grid > blocks) grid = blocks;
    if (grid > (uint32_t)a.n_ll_partials) grid = (uint32_t)a.n_ll_partials;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(em_estep_big_kernel, dim3(grid), dim3(256), smem, stream, a.xt, a.ldx, a.n, n_pad, a.D, a.params, a.K, a.lw, a.ldr,
                       a.lse, a.ll_partials);
    return (int)grid;
}

/// Returns the number of partial blocks [K][F] written, or < 0.
int launch_em_mstats_big(const MstatsArgs& a, int num_cus, hipStream_t stream)
{
    const int F = stats_count(a.d);
    if (a.mode == kFromLogRespSelfNorm) return -3;
    int splits = big_dim_splits(a.d, a.K, num_cus);
    const uint32_t n_chunks = (a.n + SC - 1) / SC;
    if ((uint32_t)splits > n_chunks) splits = (int)(n_chunks ? n_chunks : 1);
    if ((size_t)splits * a.K * F > a.partials_capacity) splits = (int)(a.partials_capacity / ((size_t)a.K * F));
    if (splits < 1) return -2;
    const uint32_t per = (n_chunks + splits - 1) / splits;
    const int T = big_tiles(a.d);
    hipLaunchKernelGGL(em_mstats_big_kernel, dim3(T * (T + 1) / 2, (a.K + KC - 1) / KC, splits), dim3(256), 0, stream, a.xt, a.ldx, a.n, a.d, a.shift, a.lw,
                       a.ldr, a.lse, a.mode, a.partials, a.K, F, per);
    return splits;
}

/// `grid_max` partial blocks of `pstride` doubles are available; the dimension-major copy of the table takes the last one (the
/// caller falls back to the plain kernel when it does not fit: fewer than ~8 clusters). Returns the partial blocks used, or 0.
int launch_kmeans_assign_big(const KmeansArgs& a, int grid_max, size_t pstride, hipStream_t stream)
{
    const int Kp = (a.K + KB - 1) / KB * KB;
    if (grid_max < 2 || (size_t)Kp * a.D > pstride) return 0;
    double* centT = a.partials + (size_t)(grid_max - 1) * pstride;
    const uint32_t n_pad = padded_samples(a.n);
    int grid = grid_max - 1;
    const uint32_t need = (n_pad + 511) / 512;
    if ((uint32_t)grid > need) grid = (int)(need ? need : 1);
    int tb = (Kp * a.D + 255) / 256;
    hipLaunchKernelGGL(kmeans_transpose_kernel, dim3(tb > 1024 ? 1024 : tb), dim3(256), 0, stream, a.centroids, a.K, Kp, a.D, centT);
    hipLaunchKernelGGL(kmeans_assign_big_kernel, dim3(grid), dim3(256), 0, stream, a.xt, a.ldx, a.n, n_pad, a.D, centT, Kp, a.K, a.labels,
                       a.old_labels, a.have_old, a.min_dist, a.partials, pstride);
    return grid;
}

}  // namespace mlhip
