// The three helpers of the reference's Benchmarks/bm_LinearAlgebra.cpp:6-48 -- xAx_symmetric, xxT, add_a_xxT at n = 4, 16, 64, 256,
// 1024 (RangeMultiplier(4)->Range(4, 1024)) -- on the HOST symbols the drop-in library exports (ml::LinearAlgebra::*,
// include/ML/LinearAlgebra.hpp). No Google Benchmark here: a plain loop that runs each case for ~0.2 s and prints ns per call.
// Built by tools/bm_clustering.py with g++ against libmlhip.so.
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "ML/LinearAlgebra.hpp"

template <class F> static double ns_per_call(F&& f)
{
    using clock = std::chrono::steady_clock;
    long reps = 1;
    for (;;) {
        const auto t0 = clock::now();
        for (long r = 0; r < reps; ++r) f();
        const double s = std::chrono::duration<double>(clock::now() - t0).count();
        if (s > 0.2) return s / (double)reps * 1e9;
        reps = s < 0.01 ? reps * 10 : (long)((double)reps * 0.25 / s) + 1;
    }
}

int main()
{
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(-1, 1);        // Eigen's Random(): uniform in [-1, 1]
    std::printf("# bm_LinearAlgebra.cpp on the host symbols of libmlhip.so (ns per call)\n");
    volatile double sink = 0;
    for (long n : {4L, 16L, 64L, 256L, 1024L}) {
        ml::MatrixXd A(n, n), dest(n, n);
        ml::VectorXd x(n);
        for (long j = 0; j < n; ++j) {
            x.data()[j] = u(rng);
            for (long i = 0; i < n; ++i) A.data()[j * n + i] = u(rng);
        }
        for (long j = 0; j < n; ++j)
            for (long i = 0; i < j; ++i) A.data()[j * n + i] = A.data()[i * n + j] = (A.data()[j * n + i] + A.data()[i * n + j]) / 2;
        for (long i = 0; i < n * n; ++i) dest.data()[i] = 0;
        const double a = ns_per_call([&] { sink = sink + ml::LinearAlgebra::xAx_symmetric(A, x); });
        const double b = ns_per_call([&] { ml::LinearAlgebra::xxT(x, dest); });
        const double c = ns_per_call([&] { ml::LinearAlgebra::add_a_xxT(x, dest, 0.5); });
        std::printf("n=%4ld: xAx_symmetric %10.1f ns, xxT %10.1f ns, add_a_xxT %10.1f ns\n", n, a, b, c);
    }
    return 0;
}
