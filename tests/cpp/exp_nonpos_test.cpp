// Host build of ml_amd/csrc/device/exp_nonpos.hpp (the same text the kernels compile): accuracy against long double expl,
// special values, monotonicity. Prints "max_ulp <x> bad <n> nonmono <n>".
#include <cmath>
#include <cstdio>
#include <random>

#include "exp_nonpos.hpp"

int main()
{
    std::mt19937_64 g(1);
    double max_ulp = 0;
    long bad = 0;
    auto test = [&](double x) {
        const long double ref = expl((long double)x);
        const double got = mlhip::exp_nonpos(x), rd = (double)ref;
        if (rd == 0 && got == 0) return;
        const double ulp = std::fabs((double)(((long double)got - ref) / (std::nextafter(rd, INFINITY) - rd)));
        if (!(ulp <= max_ulp)) max_ulp = ulp;
        if (!(ulp <= 1.0)) ++bad;
    };
    std::uniform_real_distribution<double> u(-40, 0), u2(-750, 0), u3(-1, 0);
    for (int i = 0; i < 1000000; ++i) { test(u(g)); test(u2(g)); test(u3(g)); }
    for (double x : {0.0, -0.0, -745.0, -745.13, -745.2, -746.0, -708.5, -1e300, 1e-17, -1e-17}) test(x);
    if (mlhip::exp_nonpos(0.0) != 1.0 || mlhip::exp_nonpos(-INFINITY) != 0.0 || mlhip::exp_nonpos(-746.0) != 0.0) ++bad;
    if (!std::isnan(mlhip::exp_nonpos(NAN)) || !std::isnan(mlhip::exp_nonpos(-NAN))) ++bad;   // a NaN must stay a NaN
    long nonmono = 0;
    double prev = 0;
    for (double x = -60; x < 0; x += 3e-5) { const double v = mlhip::exp_nonpos(x); if (v < prev) ++nonmono; prev = v; }
    std::printf("max_ulp %.4f bad %ld nonmono %ld\n", max_ulp, bad, nonmono);
    return (bad || nonmono) ? 1 : 0;
}
