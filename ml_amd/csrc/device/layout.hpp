// HBM / parameter / statistics layouts shared by the host code and the gfx950 kernels. No HIP types here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>

namespace mlhip {

/// Environment switches that exist only to measure a decision against its alternative (A/B runs; DESIGN.md section 7 lists them with
/// the profile that settled each): honoured by the `make EXPERIMENTS=1` library, compiled to "not set" in the default one -- the
/// shipped binary has one code path per decision and no getenv on it.
inline const char* ab_env(const char* name)
{
#ifdef MLHIP_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- layout constants shared by host and device code -------------------------------------------
constexpr int kSampleTile = 256;  // N is padded to a multiple of this in HBM
/// Samples allocated for a block of n: a whole number of tiles and at least one (an EMPTY shard of a row-sharded job still
/// runs every kernel of the iteration on one all-padding tile, so that every rank launches the same sequence).
inline uint32_t padded_samples(uint64_t n)
{
    const uint64_t p = (n + kSampleTile - 1) / kSampleTile * kSampleTile;
    return (uint32_t)(p ? p : kSampleTile);
}

/// Largest dimension the kernels are instantiated for. Up to kRegDim every kernel variant exists (the headline shapes);
/// above it only the 4x4-block matrix-core E-step, the wide statistics kernel and the matrix-core K-means kernel, with
/// fewer samples per wave in each tier (kRegDim < d <= kMidDim, kMidDim < d <= kMaxDim): a wave keeps its samples'
/// coordinates in registers.
/// Diagonal kernel: largest |(mu - shift) / sigma| for which the two-operation density form fma(a, x~, b)^2 is used
/// (em_diag.hip); beyond it the exact (x - mu)^2 / sigma^2.
constexpr double kDiagAbLimit = 64.0;
/// s_setprio level of a wave while it streams matrix instructions (0 elsewhere): see em_estep_mfma4.hip.
constexpr int kMatrixPhasePriority = 2;
constexpr int kMaxDim = 128;
constexpr int kGenericMaxDim = 4096;   // d > kMaxDim: correctness tier (device/generic_dim.hip)
constexpr int kMidDim = 64;
constexpr int kRegDim = 32;

/// Dimension the kernels are instantiated for: d is padded with zero coordinates up to this.
inline int padded_dim(int d)
{
    static const int sizes[] = {1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128};
    for (int s : sizes)
        if (d <= s) return s;
    // beyond kMaxDim: the plain kernels of generic_dim.hip (any multiple of 4; the bound keeps the packed-triangle indices in int)
    return d <= kGenericMaxDim ? (d + 3) & ~3 : -1;
}

/// E-step parameter record of one component, PS(D) doubles:
///   [ mean(D) | W = L^{-1}, lower triangle packed row by row (row j holds j+1 entries) | coef ]
/// with Sigma = L L^T and coef = log(pi) - sum_j log L_jj.
inline int estep_param_stride(int D) { return D + D * (D + 1) / 2 + 1; }

/// Matrix-core E-step (dimensions 12..32, multiples of 4): W is cut into 16-row blocks J and 4-column slabs ls; only
/// the slabs on or below the block diagonal exist. Record of one component, estep_mfma_param_stride(D) doubles:
///   [ slab c = (J, ls) in (J-major) order: 64 doubles, lane l holds W[16J + (l&15)][4ls + (l>>4)] | mean(D) | coef ].
inline int estep_mfma_slabs_of(int D, int J) { const int ls = D / 4; return 4 * (J + 1) < ls ? 4 * (J + 1) : ls; }
inline int estep_mfma_slab_count(int D) { return D <= 16 ? estep_mfma_slabs_of(D, 0) : estep_mfma_slabs_of(D, 0) + estep_mfma_slabs_of(D, 1); }
inline int estep_mfma_param_stride(int D) { return estep_mfma_slab_count(D) * 64 + D + 1; }
inline bool estep_mfma_supported(int D) { return D >= 12 && D <= 32 && D % 4 == 0; }

/// 4x4-block E-step (v_mfma_f64_4x4x4_4b_f64; dimensions 12..128, multiples of 4): W is cut into 4x4 blocks (R, C); only
/// blocks on or below the diagonal exist, ordered by column quad C, then row quad R. Record of one component,
/// estep_mfma4_param_stride(D) doubles:
///   [ block t: 16 doubles, entry [k][i] = W[4R + i][4C + k] | mean(D) | -W (mean - shift) (D) | coef ]
/// (both vectors are always there: the exact form of the kernel reads the first, the FOLD form the second).
/// FOLD form of that kernel (y = W (x - s) - W (mu - s), the second term from the record's second vector):
/// selected by the host while max_k |W_k (mu_k - s)|_inf <= kEstepFoldLimit. Each entry of y then carries an absolute error of
/// about 2^-53 * (|W (x - s)| + |W (mu - s)|) <= 2^-52 * limit = 1.4e-14 instead of a relative one, i.e. at most ~1e-13 in a
/// log-responsibility that matters (|y| of a few units, d <= 32) -- inside the 1e-12 parity tolerances, labels unaffected
/// unless two responsibilities already agree to 1e-13.
constexpr double kEstepFoldLimit = 64.0;
inline bool estep_mfma4_supported(int D) { return D >= 12 && D <= kMaxDim && D % 4 == 0; }
inline int estep_mfma4_block_count(int D) { const int q = D / 4; return q * (q + 1) / 2; }
inline int estep_mfma4_param_stride(int D) { return estep_mfma4_block_count(D) * 16 + 2 * D + 1; }

/// Sufficient statistics of one component: packed lower triangle (row-major) of sum_i r_i xt_i xt_i^T,
/// xt = [x - shift ; 1] (length d+1). Entry (a,b), a >= b, sits at a(a+1)/2 + b; so
///   S0 = (d,d), S1'_b = (d,b), M2'_ab = (a,b).
inline int stats_count(int d) { return (d + 1) * (d + 2) / 2; }
/// Diagonal-covariance extension (device/em_diag.hip). Parameter record of one component, diag_param_stride(D) doubles:
///   [ mean(D) | iv(D) = 1 / sigma_j^2 | coef = log(pi) - sum_j log sigma_j | B2 = sum_j b_j^2 ]       (even stride: 16-byte aligned reads)
/// -- what the exact density form reads (z = x - mean; q += (z iv) z). Behind the KP records of a parameter set sits a TRAILER with
/// the operands of the shift-centred forms, a = 1 / sigma, b = -(mean - shift) / sigma, dimension-major so that one dimension's
/// operands of all components are contiguous:
///   [ aT: D rows of KP doubles, aT[j KP + k] = a_kj | bT: D rows of KP, bT[j KP + k] = b_kj ]
/// B2 (NaN / inf when a parameter is not finite) is what the kernels' guards read, the same on every workgroup and rank:
///   * two-operation form  q += fma(a, x~, b)^2  (error ~ eps |b| per term): while every B2 <= kDiagAbLimit^2;
///   * expanded form on the matrix cores  lw = coef - B2/2 + sum_j (-a^2/2) x~^2 + (-a b) x~  (error ~ eps 4 B2 in q, i.e. in a
///     log-responsibility): while every B2 <= kDiagExpandLimit -- 4.5e-13, inside the 1e-12 parity tolerances.
/// Padding records k >= K: a = b = 0, coef = -inf, B2 = 0.
constexpr double kDiagExpandLimit = 1024.0;
constexpr int diag_param_stride_c(int D) { return 2 * D + 2; }
inline int diag_param_stride(int D) { return diag_param_stride_c(D); }
inline size_t diag_param_doubles(int D, int KP) { return (size_t)KP * diag_param_stride_c(D) + 2 * (size_t)D * KP; }
inline int diag_stats_count(int d) { return 2 * d + 1; }
inline int stats_index(int a, int b) { return a * (a + 1) / 2 + b; }

}  // namespace mlhip
