#include "em_math.hpp"

#include <sched.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../device/layout.hpp"
#include "team.hpp"

namespace mlhip {
namespace host {

/// L L^T = A, lower triangle, column-major. Every entry is  (A(i,j) - sum_{l<j} L(i,l) L(j,l)) / L(jj)  with the sum collected in
/// ascending l, one product and one subtraction at a time. Large d: by column PANELS with contiguous inner loops (the left-looking
/// loop walks L with stride d: 100 ms per component at d = 512); each entry still receives its terms in ascending l, so both forms
/// give the same bits.
void cholesky_lower(int d, const double* A, double* L)
{
    if (d >= 48) {
        for (int j = 0; j < d; ++j) {
            for (int i = 0; i < j; ++i) L[j * d + i] = 0.0;
            for (int i = j; i < d; ++i) L[j * d + i] = A[j * d + i];
        }
        // Panels of NB columns: (1) the panel receives the terms of all earlier columns, l ascending, one IB-row tile of it at a time
        // (the tile stays in L1 while the earlier columns stream by -- plain rank-1 updates of the whole trailing matrix move
        // d^3 / 3 doubles through memory per factorization: 15 ms at d = 512); (2) the panel is factored, right-looking inside it.
        constexpr int NB = 32, IB = 128;
        for (int j0 = 0; j0 < d; j0 += NB) {
            const int j1 = j0 + NB < d ? j0 + NB : d;
            for (int i0 = j0; i0 < d; i0 += IB) {
                const int i1 = i0 + IB < d ? i0 + IB : d;
                for (int l = 0; l < j0; ++l) {
                    const double* cl = L + (size_t)l * d;
                    for (int j = j0; j < j1; ++j) {
                        double* cj = L + (size_t)j * d;
                        const double ljl = cl[j];
                        for (int i = (i0 > j ? i0 : j); i < i1; ++i) cj[i] -= cl[i] * ljl;
                    }
                }
            }
            for (int l = j0; l < j1; ++l) {
                double* cl = L + (size_t)l * d;
                const double lll = std::sqrt(cl[l]);
                cl[l] = lll;
                for (int i = l + 1; i < d; ++i) cl[i] = cl[i] / lll;
                for (int j = l + 1; j < j1; ++j) {
                    double* cj = L + (size_t)j * d;
                    const double ljl = cl[j];
                    for (int i = j; i < d; ++i) cj[i] -= cl[i] * ljl;
                }
            }
        }
        return;
    }
    for (int i = 0; i < d * d; ++i) L[i] = 0.0;
    for (int j = 0; j < d; ++j) {
        double s = A[j * d + j];
        for (int l = 0; l < j; ++l) s -= L[l * d + j] * L[l * d + j];
        const double ljj = std::sqrt(s);
        L[j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            double t = A[j * d + i];
            for (int l = 0; l < j; ++l) t -= L[l * d + i] * L[l * d + j];
            L[j * d + i] = t / ljj;
        }
    }
}

void process_covariance(int d, const double* cov, double* inverse, double* sqrt_det)
{
    std::vector<double> L((size_t)d * d), y(d);
    cholesky_lower(d, cov, L.data());
    for (int c = 0; c < d; ++c) {
        for (int i = 0; i < d; ++i) {
            double t = (i == c) ? 1.0 : 0.0;
            for (int l = 0; l < i; ++l) t -= L[l * d + i] * y[l];
            y[i] = t / L[i * d + i];
        }
        for (int i = d - 1; i >= 0; --i) {
            double t = y[i];
            for (int l = i + 1; l < d; ++l) t -= L[i * d + l] * inverse[c * d + l];
            inverse[c * d + i] = t / L[i * d + i];
        }
    }
    double sd = 1.0;
    for (int i = 0; i < d; ++i) sd *= L[i * d + i];
    *sqrt_det = sd;
}

namespace {
/// Threads used for the K independent per-component d x d factorizations (serial section of an EM iteration: every
/// microsecond here is paid by all GPUs of a row-sharded job). At most 8, and never more than this process's share of
/// the host cores when several ranks run on one node (set_host_ranks): oversubscribed teams only take turns on the same cores.
std::atomic<int> g_host_threads{0};

int resolve_host_threads(int local_ranks)
{
    if (const char* e = std::getenv("MLHIP_HOST_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1) return v > 64 ? 64 : v;
    }
    int cores = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
    if (local_ranks < 1) local_ranks = 1;
    int t = cores / local_ranks;
    return t < 1 ? 1 : (t > 8 ? 8 : t);
}

/// A thread team costs 10-20 us to wake: only worth it when the K factorizations (~2.3 d^3 flop each) are more than that.
inline bool worth_threads(int K, int d) { return (double)K * d * d * d >= 4.0e5; }

int host_threads()
{
    int t = g_host_threads.load(std::memory_order_relaxed);
    if (t == 0) {
        t = resolve_host_threads(1);
        g_host_threads.store(t, std::memory_order_relaxed);
    }
    return t;
}

/// fn(k, L, W) for the K components, L and W two d x d scratch matrices of the thread that runs k: on this thread's team
/// (host/team.hpp) when the K factorizations are worth waking it, else a plain loop. Every k is handled by exactly one thread and
/// touches only its own outputs: the results do not depend on the number of threads.
template <class Fn> void for_each_component(int K, int d, Fn&& fn)
{
    const int threads = worth_threads(K, d) ? host_threads() : 1;
    if (threads <= 1 || K <= 1) {
        std::vector<double> L((size_t)d * d), W((size_t)d * d);
        for (int k = 0; k < K; ++k) fn(k, L, W);
        return;
    }
    const std::function<void(int)> body = [&](int k) {
        thread_local std::vector<double> L, W;                 // (grown once per worker, reused by every region)
        if (L.size() < (size_t)d * d) { L.resize((size_t)d * d); W.resize((size_t)d * d); }
        fn(k, L, W);
    };
    Team::mine().for_each(K, threads, body);
}

/// W = L^-1 (lower triangular, column-major d x d) and sum_j log L_jj for one covariance.
double whitening_matrix(int d, const double* cov, std::vector<double>& L, std::vector<double>& W)
{
    cholesky_lower(d, cov, L.data());
    if (d >= 48) {
        // column c of W = L^-1 by forward substitution, right-looking: w_l is final once the terms of all l' < l are in, and its
        // product goes to every later entry at once, contiguous in i; the terms of an entry still arrive in ascending l (same bits
        // as the loop below, which walks L with stride d)
        // ... NB columns of W side by side, so that a column of L is read once for all of them (one column at a time reads the whole
        // of L per column of W: d^3 / 2 doubles per inverse)
        constexpr int NB = 32;
        for (int c0 = 0; c0 < d; c0 += NB) {
            const int c1 = c0 + NB < d ? c0 + NB : d;
            for (int c = c0; c < c1; ++c) {
                double* w = W.data() + (size_t)c * d;
                for (int i = 0; i < d; ++i) w[i] = (i == c) ? 1.0 : 0.0;
            }
            for (int l = c0; l < d; ++l) {
                const double* cl = L.data() + (size_t)l * d;
                const int c_last = l < c1 - 1 ? l : c1 - 1;             // columns c <= l of the block have an entry l
                for (int c = c0; c <= c_last; ++c) {
                    double* w = W.data() + (size_t)c * d;
                    const double wl = w[l] / cl[l];
                    w[l] = wl;
                    for (int i = l + 1; i < d; ++i) w[i] -= cl[i] * wl;
                }
            }
        }
    } else {
    for (int c = 0; c < d; ++c) {
        for (int i = 0; i < d; ++i) {
            if (i < c) { W[c * d + i] = 0.0; continue; }
            double t = (i == c) ? 1.0 : 0.0;
            for (int l = c; l < i; ++l) t -= L[l * d + i] * W[c * d + l];
            W[c * d + i] = t / L[i * d + i];
        }
    }
    }
    double log_det_half = 0.0;
    for (int j = 0; j < d; ++j) log_det_half += std::log(L[j * d + j]);
    return log_det_half;
}
}  // namespace

#ifdef MLHIP_EXPERIMENTS
void build_estep_params_mfma(int d, int D, int K, const double* mixing, const double* means, const double* covariances,
                             double* records)
{
    const int PS = estep_mfma_param_stride(D);
    const int NC = estep_mfma_slab_count(D);
    const int JB = (D + 15) / 16;
    std::vector<double> L((size_t)d * d), W((size_t)d * d);
    for (int k = 0; k < K; ++k) {
        double* rec = records + (size_t)k * PS;
        for (int i = 0; i < PS; ++i) rec[i] = 0.0;
        const double log_det_half = whitening_matrix(d, covariances + (size_t)k * d * d, L, W);
        int c = 0;
        for (int J = 0; J < JB; ++J)
            for (int ls = 0; ls < estep_mfma_slabs_of(D, J); ++ls, ++c)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = 16 * J + (lane & 15), col = 4 * ls + (lane >> 4);
                    rec[c * 64 + lane] = (row < d && col <= row) ? W[col * d + row] : 0.0;
                }
        for (int j = 0; j < d; ++j) rec[NC * 64 + j] = means[(size_t)k * d + j];
        rec[NC * 64 + D] = std::log(mixing[k]) - log_det_half;
    }
}

#endif

bool build_estep_params_mfma4(int d, int D, int K, const double* mixing, const double* means, const double* covariances,
                              const double* shift, double fold_limit, double* records)
{
    const int PS = estep_mfma4_param_stride(D);
    const int NB = estep_mfma4_block_count(D);
    const int Q = D / 4;
    std::vector<double> folded(shift ? (size_t)K * d : 0);     // c_k = W_k (mu_k - shift)
    std::vector<double> biggest_of((size_t)K, 0.0);            // per component; the maximum below is order-independent
    for_each_component(K, d, [&](int k, std::vector<double>& L, std::vector<double>& W) {
        double* rec = records + (size_t)k * PS;
        for (int i = 0; i < PS; ++i) rec[i] = 0.0;
        const double log_det_half = whitening_matrix(d, covariances + (size_t)k * d * d, L, W);
        int t = 0;
        for (int C = 0; C < Q; ++C)
            for (int R = C; R < Q; ++R, ++t)
                for (int kk = 0; kk < 4; ++kk)
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * R + i, col = 4 * C + kk;
                        rec[t * 16 + kk * 4 + i] = (row < d && col <= row) ? W[col * d + row] : 0.0;
                    }
        for (int j = 0; j < d; ++j) rec[NB * 16 + j] = means[(size_t)k * d + j];
        rec[NB * 16 + 2 * D] = std::log(mixing[k]) - log_det_half;
        if (shift) {
            double biggest = 0.0;
            for (int row = 0; row < d; ++row) {
                double c = 0.0;
                for (int col = 0; col <= row; ++col) c += W[col * d + row] * (means[(size_t)k * d + col] - shift[col]);
                folded[(size_t)k * d + row] = c;
                const double a = std::fabs(c);
                if (a > biggest) biggest = a;
            }
            biggest_of[k] = biggest;
        }
    });
    double biggest = 0.0;
    for (int k = 0; k < K; ++k) biggest = biggest_of[k] > biggest ? biggest_of[k] : biggest;
    if (!shift) return false;
    // the accumulator initialiser of the FOLD form (sign flipped); usable while every entry is small and finite
    bool finite = true;
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < d; ++j) {
            records[(size_t)k * PS + NB * 16 + D + j] = -folded[(size_t)k * d + j];
            finite = finite && std::isfinite(folded[(size_t)k * d + j]);
        }
    return finite && biggest <= fold_limit;                // broken parameters: leave them to the exact form (NaN in, NaN out)
}

void build_estep_params(int d, int D, int K, const double* mixing, const double* means, const double* covariances,
                        double* records)
{
    const size_t PS = (size_t)D + (size_t)D * (D + 1) / 2 + 1;       // estep_param_stride(D)
    // (this layout also serves d > 128, where a factorization is milliseconds: the K of them on the host's threads, like the
    // builders above)
    for_each_component(K, d, [&](int k, std::vector<double>& L, std::vector<double>& W) {
        double* rec = records + (size_t)k * PS;
        for (size_t i = 0; i < PS; ++i) rec[i] = 0.0;
        for (int j = 0; j < d; ++j) rec[j] = means[(size_t)k * d + j];
        const double log_det_half = whitening_matrix(d, covariances + (size_t)k * d * d, L, W);
        double* w = rec + D;
        for (int l = 0; l < d; ++l)                                   // (column l of W is contiguous: read it once, scatter into the rows)
            for (int j = l; j < d; ++j) w[(size_t)j * (j + 1) / 2 + l] = W[(size_t)l * d + j];
        rec[PS - 1] = std::log(mixing[k]) - log_det_half;
    });
}

void finalize_mstep(int d, int K, const double* stats, const double* shift, double n_global, double* mixing,
                    double* means, double* covariances)
{
    const int F = stats_count(d);
    for_each_component(K, d, [&](int k, std::vector<double>& m, std::vector<double>&) {   // (m: d of the d x d scratch doubles)
        const double* s = stats + (size_t)k * F;
        const double s0 = s[stats_index(d, d)];
        for (int a = 0; a < d; ++a) {
            m[a] = s[stats_index(d, a)] / s0;
            means[(size_t)k * d + a] = shift[a] + m[a];
        }
        double* cov = covariances + (size_t)k * d * d;
        for (int a = 0; a < d; ++a)
            for (int b = 0; b <= a; ++b) {
                const double v = (s[stats_index(a, b)] - s[stats_index(d, a)] * m[b]) / s0;
                cov[b * d + a] = v;
                cov[a * d + b] = v;
            }
        static constexpr double epsilon = 1e-15;   // ML/EM.cpp:252
        for (int a = 0; a < d; ++a) cov[a * d + a] += epsilon;
        mixing[k] = s0 / n_global;                 // ML/EM.cpp:257
    });
}

void build_diag_params(int d, int D, int K, int K_padded, const double* mixing, const double* means, const double* variances,
                       const double* shift, double* records)
{
    const int PS = diag_param_stride(D), KP = K_padded;
    double* aT = records + (size_t)KP * PS;                    // trailer (layout.hpp): operands of the two-operation density form
    double* bT = aT + (size_t)D * KP;
    for (size_t i = 0; i < 2 * (size_t)D * KP; ++i) aT[i] = 0.0;
    for (int k = K; k < K_padded; ++k) {
        double* rec = records + (size_t)k * PS;
        for (int i = 0; i < PS; ++i) rec[i] = 0.0;
        rec[2 * D] = -HUGE_VAL;                                // (a = b = 0, B2 = 0: every density form gives lw = -inf)
    }
    for (int k = 0; k < K; ++k) {
        double* rec = records + (size_t)k * PS;
        for (int i = 0; i < PS; ++i) rec[i] = 0.0;            // padded coordinates: mean 0, weight 0 -> contribute exactly 0
        double log_det_half = 0.0;
        double b2 = 0.0;
        for (int j = 0; j < d; ++j) {
            const double l = std::sqrt(variances[(size_t)k * d + j]);
            const double mean = means[(size_t)k * d + j];
            rec[j] = mean;
            rec[D + j] = (1.0 / l) / l;
            log_det_half += std::log(l);
            // (the device closing kernel writes the same: em_close.hip)
            const double a = 1.0 / l;
            const double b = -((mean - shift[j]) * a);
            aT[(size_t)j * KP + k] = a;
            bT[(size_t)j * KP + k] = b;
            b2 = std::fma(b, b, b2);                          // (em_close.hip sums in the same order with the same fma)
            if (!std::isfinite(a)) b2 = HUGE_VAL;
        }
        rec[2 * D] = std::log(mixing[k]) - log_det_half;
        rec[2 * D + 1] = b2;
    }
}

void finalize_mstep_diag(int d, int K, const double* stats, const double* shift, double n_global, double* mixing,
                         double* means, double* variances)
{
    const int F = diag_stats_count(d);
    for (int k = 0; k < K; ++k) {
        const double* s = stats + (size_t)k * F;
        const double s0 = s[2 * d];
        for (int a = 0; a < d; ++a) {
            const double m = s[a] / s0;
            means[(size_t)k * d + a] = shift[a] + m;
            variances[(size_t)k * d + a] = (s[d + a] - s[a] * m) / s0 + 1e-15;   // ridge: ML/EM.cpp:252
        }
        mixing[k] = s0 / n_global;                                                // ML/EM.cpp:257
    }
}

void set_host_ranks(int local_ranks) { g_host_threads.store(resolve_host_threads(local_ranks), std::memory_order_relaxed); }

}  // namespace host
}  // namespace mlhip
