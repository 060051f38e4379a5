// Exact fixed-point accumulation helpers shared by the K-means kernels (kmeans.hip, kmeans_mfma.hip).
#pragma once

namespace mlhip {

typedef unsigned long long u64;

/// t (|t| < 2^94, only the integer part is kept) -> limbs: t = i2 * 2^64 + u1 * 2^32 + u0 (+ dropped fraction), with
/// u0, u1 in [0, 2^32). The limbs of many samples are summed with 64-bit INTEGER atomics: integer addition is associative,
/// so the sums do not depend on the order in which lanes, waves, workgroups or GPUs contribute.
__device__ __forceinline__ void split_limbs(double t, u64& w0, u64& w1, u64& w2)
{
    const double h2 = floor(t * 0x1p-64);
    const double r = __builtin_fma(-h2, 0x1p64, t);         // exact, in [0, 2^64)
    const double h1 = floor(r * 0x1p-32);
    const double l = __builtin_fma(-h1, 0x1p32, r);          // exact, in [0, 2^32)
    w2 = (u64)(long long)(int)h2;                            // |h2| < 2^30
    w1 = (u64)(unsigned)h1;
    w0 = (u64)(unsigned)l;                                   // truncates the fraction below one unit
}

}  // namespace mlhip
