// The vector-unit form of the fused E-step + statistics pass (few components in few dimensions: the reference's own benchmark
// regime, d = 2, K = 3, Benchmarks/bm_EM.cpp) as device functions, shared by the one-pass kernel of em_fused_small.hip and by the
// device-resident loop of em_resident.hip -- one text, so the two give the same bits.
//
// What it computes per sample i and component k (reference ML/EM.cpp:190-219, 221-250):
//   lw_ik = log pi_k - sum log L_jj - |W_k (x_i - mu_k)|^2 / 2,   r_ik = exp(lw_ik - lse_i),   lse_i = log sum_k exp(lw_ik)
//   acc[k][f] += r_ik phi_f(x~_i),   phi = vech([x~ ; 1][x~ ; 1]^T),   x~ = x - shift      (K F fused multiply-adds per sample)
// Every lane keeps the K F accumulators of its samples in registers; the lanes are summed ONCE, at the end of a workgroup's pass
// (valu_fold: halving exchanges over the wave, then the four waves in order through LDS): fixed order, reproducible.
#pragma once
#include "em_mstats_common.hpp"
#include "exp_nonpos.hpp"
#include "lane_ops.hpp"

namespace mlhip {
namespace mstats {

/// (a, b) -> one value per lane: the lower half of the lanes (of the wave: BIT5; of every 32 lanes: the even 16-lane row) gets
/// a_l + a_partner, the upper half b_partner + b_l, partner = l ^ 32 (l ^ 16). v_permlane32_swap (v_permlane16_swap) exchanges the
/// upper half of its first operand with the lower half of its second; no LDS traffic.
template <bool BIT5> __device__ __forceinline__ double halves_fold(double a, double b)
{
    const unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
    if constexpr (BIT5) {
        const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
}

template <int D, int K> struct ValuShape {
    static constexpr int PS = D + D * (D + 1) / 2 + 1;            // estep_param_stride(D)
    static constexpr int DA = D + 1, F = DA * (DA + 1) / 2;
    static constexpr int V = K * F, VP = (V + 3) / 4 * 4;         // accumulators per lane, padded for the two halving steps of valu_fold
};

/// The samples of tile `tile` of this lane -> x (tiles are whole inside the allocation: i < n_pad).
template <int D> __device__ __forceinline__ void valu_load_tile(const double* __restrict__ xt, size_t ldx, uint32_t tile, int lane, double (&x)[D])
{
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = xt[(size_t)j * ldx + (size_t)tile * TS + lane];
}

/// One wave's pass over its tiles block * 4 + wave, + 4 * grid, ... of the (virtual) workgroup `block` of `grid`. `xn` holds the
/// first of those tiles on entry (valu_load_tile; any tile when the wave has none) and is left holding the wave's LAST tile again.
/// `rec`: the K records [mean(D) | W lower triangle, row by row | coef] -- a wave-uniform global pointer (scalar loads, as in
/// em_estep.hip) or an LDS pointer (broadcast reads): the same arithmetic on the same values either way.
/// lse_out may be null (the resident loop: whoever needs lse rebuilds it with the log-responsibilities, ensure_lw).
template <int D, int K, typename P, typename Probe = NoProbe>
__device__ __forceinline__ void valu_tiles(const double* __restrict__ xt, size_t ldx, uint32_t n, const double* __restrict__ shift, P rec,
                                           double* __restrict__ lse_out, uint32_t block, uint32_t grid, int wave, int lane,
                                           double (&xn)[D], double (&acc)[ValuShape<D, K>::VP], double& ll_acc, const Probe& probe = Probe())
{
#pragma clang fp contract(off)     // every fused multiply-add below is written out: the one-pass kernel and the resident loop instantiate
                                   // this text with different record pointers and must not be contracted differently
    using S = ValuShape<D, K>;
    constexpr int PS = S::PS, DA = S::DA, F = S::F;
    const uint32_t n_tiles = (n + TS - 1) / TS;
    const uint32_t stride = grid * 4;
    for (uint32_t tile = block * 4 + wave; tile < n_tiles; tile += stride) {
        const uint32_t i = tile * TS + lane;                  // < n_pad: inside the allocation
        const bool live = i < n;
        double x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = xn[j];
        valu_load_tile<D>(xt, ldx, tile + stride < n_tiles ? tile + stride : tile, lane, xn);   // in flight while this one is worked on
        // ---- log-densities (em_estep.hip's arithmetic: z = x - mu, y = W z, lw = coef - |y|^2 / 2)
        double lwv[K];
        double m = -__builtin_inf();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            {
                P p = rec + (size_t)k * PS;
                double z[D];
#pragma unroll
                for (int j = 0; j < D; ++j) z[j] = x[j] - p[j];
                P w = p + D;
                double q = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double y = w[j * (j + 1) / 2] * z[0];
#pragma unroll
                    for (int l = 1; l <= j; ++l) y = __builtin_fma(w[j * (j + 1) / 2 + l], z[l], y);
                    q = __builtin_fma(y, y, q);
                }
                const double lw = __builtin_fma(-0.5, q, p[PS - 1]);
                lwv[k] = lw;
                m = lw > m ? lw : m;
            }
        }
        probe(8);
        // ---- normalisation: one exp per (sample, component), the K polynomial chains side by side (exp_nonpos.hpp)
#pragma unroll
        for (int k = 0; k < K; ++k) lwv[k] -= m;
        exp_nonpos_n<K>(lwv);
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += lwv[k];
        const double lse = m + log(s);
        if (lse_out) lse_out[i] = lse;
        if (live) ll_acc += lse;
        const double inv = live ? 1.0 / s : 0.0;              // padding samples contribute nothing
        probe(9);
        // ---- statistics: phi_(a, b) = x~_a x~_b, a >= b, x~ = [x - shift ; 1], packed at a (a + 1) / 2 + b
        double xs[DA];
#pragma unroll
        for (int j = 0; j < D; ++j) xs[j] = x[j] - shift[j];
        xs[D] = 1.0;
        double phi[F];
#pragma unroll
        for (int a = 0; a < DA; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) phi[a * (a + 1) / 2 + b] = a == D ? xs[b] : xs[a] * xs[b];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double r = lwv[k] * inv;
#pragma unroll
            for (int f = 0; f < F; ++f) acc[k * F + f] = __builtin_fma(r, phi[f], acc[k * F + f]);
        }
        probe(10);
    }
}

/// The 64 lanes of every accumulator, summed in a fixed order, into row `wave` of fold[4][VP] (and the wave's log-likelihood sum
/// into red[wave]). The first two steps HALVE the values a lane holds: lanes 0-31 take accumulators [0, VP / 2) of both halves of
/// the wave, lanes 32-63 the rest (one v_permlane32_swap per word and one addition per PAIR), then the same between the 16-lane
/// rows; what is left is VP / 4 values over 16 lanes, summed towards lane 0 of the row in four steps  v_l += v_(l + 8), (l + 4),
/// (l + 2), (l + 1)  -- the tree of the xor butterfly this replaces (round 4: ds_bpermute, one dependent LDS round trip per step and
/// value: 2.3 of the 3.5 us a one-tile pass took at K F = 45), every step issued for ALL values before the next. The caller's
/// workgroup barrier comes next; entry e of the workgroup is then ((fold[0][e] + fold[1][e]) + fold[2][e]) + fold[3][e].
template <int VP> __device__ __forceinline__ void valu_fold(double (&acc)[VP], double ll_acc, int wave, int lane, double (*fold)[VP], double* red)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < VP / 2; ++e) acc[e] = halves_fold<true>(acc[e], acc[e + VP / 2]);
#pragma unroll
    for (int e = 0; e < VP / 4; ++e) acc[e] = halves_fold<false>(acc[e], acc[e + VP / 4]);
#pragma unroll
    for (int e = 0; e < VP / 4; ++e) acc[e] += row_shift_left<8>(acc[e]);
#pragma unroll
    for (int e = 0; e < VP / 4; ++e) acc[e] += row_shift_left<4>(acc[e]);
#pragma unroll
    for (int e = 0; e < VP / 4; ++e) acc[e] += row_shift_left<2>(acc[e]);
#pragma unroll
    for (int e = 0; e < VP / 4; ++e) acc[e] += row_shift_left<1>(acc[e]);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int e = 0; e < VP / 4; ++e) fold[wave][e + (lane >> 4) * (VP / 4)] = acc[e];   // row r of the wave holds accumulators r VP / 4 + e
    }
    // the wave's log-likelihood sum: lane 0 of  v_l += v_(l + 32), (l + 16), (l + 8), ..., (l + 1)
    ll_acc += upper_half<true>(ll_acc);
    ll_acc += upper_half<false>(ll_acc);
    ll_acc += row_shift_left<8>(ll_acc);
    ll_acc += row_shift_left<4>(ll_acc);
    ll_acc += row_shift_left<2>(ll_acc);
    ll_acc += row_shift_left<1>(ll_acc);
    if (lane == 0) red[wave] = ll_acc;
}

/// Largest K the vector-unit form is built for at dimension d (K F <= ~100 accumulators per lane); 0: not built for d.
constexpr int valu_max_k(int D) { return D == 1 ? 32 : D == 2 ? 16 : D == 3 ? 10 : D == 4 ? 7 : D == 6 ? 4 : 0; }

}  // namespace mstats
}  // namespace mlhip
