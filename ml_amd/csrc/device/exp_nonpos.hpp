// exp(x) for x <= 0 (differences lw - max, lw - lse): the only exponentials the EM kernels need. 20 fp64 operations, no
// table, no branches, few live registers -- the library exp costs ~28 and, inlined several times next to the statistics
// kernel's 160 accumulator registers, pushed it into spills. Cody-Waite reduction x = n ln2 + r, |r| <= ln2 / 2, degree-13
// Taylor polynomial (truncation 0.3466^14 / 14! = 4e-18 relative), scaling by ldexp (correct gradual underflow; exactly 0
// below -745.2 like the library function; NaN for NaN). Measured against the correctly rounded result: <= 1 ulp
// (tests/test_exp_nonpos.py). Plain C++: the same text compiles for the host in that test.
#pragma once
#include <cmath>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define MLHIP_EXP_FN __device__ __host__ __forceinline__
#else
#define MLHIP_EXP_FN inline
#endif

namespace mlhip {

MLHIP_EXP_FN double exp_nonpos(double x)
{
    x = x < -800.0 ? -800.0 : x;                                   // -inf (lw of a padding row minus the max) -> 0 below; a NaN stays a NaN: a
                                                                   // component with NaN parameters poisons the sample's sum and the log-likelihood, as in
                                                                   // the reference (ML/EM.cpp:205-218) -- the fit then never reports convergence
    const double n = __builtin_rint(x * 1.4426950408889634074);    // round(x / ln 2)
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);   // ln2_hi: low 32 mantissa bits zero, n * ln2_hi is exact
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);          // ln2_lo
    double p = 1.6059043836821614599e-10;                          // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868098979e-09);            // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441718775e-08);            // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985890653e-07);            // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985892511e-06);            // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587301566e-05);            // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841253e-04);            // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889419e-03);            // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332177e-03);            // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664354e-02);            // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);            // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

/// N exponentials side by side: the same operations as exp_nonpos on each element (bit-identical results), issued step by step
/// across the N arguments -- the degree-13 Horner chain of ONE argument is 15 dependent fp64 instructions, each waiting out the
/// pipeline latency of its predecessor; interleaved, the N chains fill each other's latency slots (em_diag.hip: the compiler
/// otherwise evaluates the 16 exponentials of a sample one after the other).
template <int N> MLHIP_EXP_FN void exp_nonpos_n(double (&x)[N])
{
    double n[N], r[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] < -800.0 ? -800.0 : x[i];
#pragma unroll
    for (int i = 0; i < N; ++i) n[i] = __builtin_rint(x[i] * 1.4426950408889634074);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(n[i], -6.93147180369123816490e-01, x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(n[i], -1.90821492927058770002e-10, r[i]);
    constexpr double c[13] = {2.0876756987868098979e-09, 2.5052108385441718775e-08, 2.7557319223985890653e-07,
                              2.7557319223985892511e-06, 2.4801587301587301566e-05, 1.9841269841269841253e-04,
                              1.3888888888888889419e-03, 8.3333333333333332177e-03, 4.1666666666666664354e-02,
                              1.6666666666666665741e-01, 0.5, 1.0, 1.0};
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = 1.6059043836821614599e-10;
#pragma unroll
    for (int t = 0; t < 13; ++t)
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = __builtin_fma(p[i], r[i], c[t]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = __builtin_ldexp(p[i], (int)n[i]);
}

}  // namespace mlhip
