// TEST INFRASTRUCTURE ONLY -- flat C entry points over the CPU restatement, for ctypes.
// See oracle/README.md. Nothing under ml_amd/ or include/ may link this.
#include "mlpp_oracle.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

using namespace oracle;

namespace {
thread_local std::string g_err;

template <class F> int guarded(F&& f)
{
    try { f(); return 0; }
    catch (const std::invalid_argument& e) { g_err = e.what(); return -1; }
    catch (const std::domain_error& e) { g_err = e.what(); return -2; }
    catch (const std::exception& e) { g_err = e.what(); return -3; }
}

std::shared_ptr<const CentroidsInitialiser> make_init(int kind, const double* fixed, unsigned d, unsigned K)
{
    switch (kind) {
    case 0: return std::make_shared<Forgy>();
    case 1: return std::make_shared<RandomPartition>();
    case 2: return std::make_shared<KPP>();
    case 3: {
        auto p = std::make_shared<FixedCentroids>();
        p->c = Mat(d, K);
        std::copy_n(fixed, static_cast<std::size_t>(d) * K, p->c.a.data());
        return p;
    }
    default: throw std::invalid_argument("unknown initialiser kind");
    }
}
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- LinearAlgebra -------------------------------------------------------------------------------
int orc_xAx_symmetric(const double* A, unsigned rows, unsigned cols, const double* x, unsigned xlen, double* out)
{
    return guarded([&] {
        Mat m(rows, cols);
        std::copy_n(A, static_cast<std::size_t>(rows) * cols, m.a.data());
        *out = xAx_symmetric(m, x, xlen);
    });
}
int orc_xxT(const double* x, unsigned n, double* dest)
{
    return guarded([&] {
        Mat m;
        xxT(x, n, m);
        std::copy(m.a.begin(), m.a.end(), dest);
    });
}
int orc_add_a_xxT(const double* x, unsigned n, double* dest, unsigned drows, unsigned dcols, double a)
{
    return guarded([&] {
        Mat m(drows, dcols);
        std::copy_n(dest, static_cast<std::size_t>(drows) * dcols, m.a.data());
        add_a_xxT(x, n, m, a);
        std::copy(m.a.begin(), m.a.end(), dest);
    });
}

int orc_calculate_XXt_beta(const double* X, unsigned q, unsigned n, const double* y, unsigned ylen, const double* lambda,
                           unsigned lambda_len, double* XXt, double* beta)
{
    return guarded([&] {
        Mat m;
        const auto b = calculate_XXt_beta(DataView{X, q, n, q}, y, ylen, m, lambda, lambda_len);
        std::copy(m.a.begin(), m.a.end(), XXt);
        std::copy(b.begin(), b.end(), beta);
    });
}

// ---- initialisers (seeded std::default_random_engine, like a freshly seeded model) ----------------
// kind: 0 Forgy, 1 RandomPartition, 2 KPP. seed_set==0 -> default-constructed engine.
int orc_init_centroids(int kind, const double* x, unsigned d, unsigned n, unsigned ld, unsigned K,
                       int seed_set, unsigned seed, double* centroids)
{
    return guarded([&] {
        std::default_random_engine prng;
        if (seed_set) prng.seed(seed);
        Mat c(d, K);
        make_init(kind, nullptr, d, K)->init(DataView{x, d, n, ld}, prng, K, c);
        std::copy(c.a.begin(), c.a.end(), centroids);
    });
}
int orc_init_closest_centroid(int kind, const double* x, unsigned d, unsigned n, unsigned ld, unsigned K,
                              int seed_set, unsigned seed, double* resp)
{
    return guarded([&] {
        std::default_random_engine prng;
        if (seed_set) prng.seed(seed);
        Mat r(n, K);
        ClosestCentroid(make_init(kind, nullptr, d, K)).init(DataView{x, d, n, ld}, prng, K, r);
        std::copy(r.a.begin(), r.a.end(), resp);
    });
}

// ---- EM ---------------------------------------------------------------------------------------------
struct OrcEM { EM em; unsigned d = 0; explicit OrcEM(unsigned K) : em(K) {} };

int orc_em_create(unsigned K, OrcEM** out) { return guarded([&] { *out = new OrcEM(K); }); }
void orc_em_destroy(OrcEM* h) { delete h; }
int orc_em_set_seed(OrcEM* h, unsigned s) { return guarded([&] { h->em.set_seed(s); }); }
int orc_em_set_absolute_tolerance(OrcEM* h, double t) { return guarded([&] { h->em.set_absolute_tolerance(t); }); }
int orc_em_set_relative_tolerance(OrcEM* h, double t) { return guarded([&] { h->em.set_relative_tolerance(t); }); }
int orc_em_set_maximum_steps(OrcEM* h, unsigned m) { return guarded([&] { h->em.set_maximum_steps(m); }); }
int orc_em_set_maximise_first(OrcEM* h, int b) { return guarded([&] { h->em.set_maximise_first(b != 0); }); }
// kind 0..2 as above, 3 = fixed centroids (d x K column-major in `fixed`).
int orc_em_set_diagonal(OrcEM* h, int on)
{
    return guarded([&] { h->em.set_diagonal(on != 0); });
}
int orc_em_set_means_initialiser(OrcEM* h, int kind, const double* fixed, unsigned d)
{
    return guarded([&] { h->em.set_means_initialiser(make_init(kind, fixed, d, h->em.K())); });
}
int orc_em_set_responsibilities_initialiser(OrcEM* h, int kind, const double* fixed, unsigned d)
{
    return guarded([&] {
        h->em.set_responsibilities_initialiser(std::make_shared<ClosestCentroid>(make_init(kind, fixed, d, h->em.K())));
    });
}
int orc_em_fit(OrcEM* h, const double* x, unsigned d, unsigned n, unsigned ld, int* converged)
{
    return guarded([&] {
        h->d = d;
        *converged = h->em.fit(DataView{x, d, n, ld}) ? 1 : 0;
    });
}
// Explicit single steps: set parameters, then E and/or M.
int orc_em_set_parameters(OrcEM* h, unsigned d, const double* means, const double* covs, const double* pis)
{
    return guarded([&] {
        const unsigned K = h->em.K();
        h->d = d;
        Mat m(d, K);
        std::copy_n(means, static_cast<std::size_t>(d) * K, m.a.data());
        std::vector<Mat> c(K, Mat(d, d));
        for (unsigned k = 0; k < K; ++k) std::copy_n(covs + static_cast<std::size_t>(k) * d * d, static_cast<std::size_t>(d) * d, c[k].a.data());
        h->em.set_parameters(m, c, std::vector<double>(pis, pis + K));
    });
}
int orc_em_expectation_step(OrcEM* h, const double* x, unsigned d, unsigned n, unsigned ld)
{
    return guarded([&] {
        h->em.prepare_for_steps(DataView{x, d, n, ld});
        h->em.expectation_step(DataView{x, d, n, ld});
    });
}
int orc_em_maximisation_step(OrcEM* h, const double* x, unsigned d, unsigned n, unsigned ld)
{
    return guarded([&] { h->em.maximisation_step(DataView{x, d, n, ld}); });
}
int orc_em_calculate_labels(OrcEM* h) { return guarded([&] { h->em.calculate_labels(); }); }
// Overwrite the stored responsibilities (n x K column-major), e.g. to run an M-step from given R.
int orc_em_set_responsibilities(OrcEM* h, const double* r, unsigned d, unsigned n)
{
    return guarded([&] {
        const double dummy = 0;
        h->em.prepare_for_steps(DataView{&dummy, d, n, d});
        std::copy_n(r, static_cast<std::size_t>(n) * h->em.K(), h->em.mutable_responsibilities().a.data());
    });
}
int orc_em_assign_responsibilities(OrcEM* h, const double* x, unsigned xlen, double* u, unsigned ulen)
{
    return guarded([&] { h->em.assign_responsibilities(x, xlen, u, ulen); });
}
double orc_em_log_likelihood(const OrcEM* h) { return h->em.log_likelihood(); }
int orc_em_converged(const OrcEM* h) { return h->em.converged() ? 1 : 0; }
unsigned orc_em_steps_done(const OrcEM* h) { return h->em.steps_done(); }
void orc_em_get_means(const OrcEM* h, double* out) { const auto& m = h->em.means(); std::copy(m.a.begin(), m.a.end(), out); }
void orc_em_get_mixing_probabilities(const OrcEM* h, double* out) { const auto& p = h->em.mixing_probabilities(); std::copy(p.begin(), p.end(), out); }
void orc_em_get_covariances(const OrcEM* h, double* out)
{
    for (const Mat& c : h->em.covariances()) out = std::copy(c.a.begin(), c.a.end(), out);
}
void orc_em_get_inverse_covariances(const OrcEM* h, double* out)
{
    for (const Mat& c : h->em.inverse_covariances()) out = std::copy(c.a.begin(), c.a.end(), out);
}
void orc_em_get_sqrt_dets(const OrcEM* h, double* out) { const auto& p = h->em.sqrt_dets(); std::copy(p.begin(), p.end(), out); }
void orc_em_get_responsibilities(const OrcEM* h, double* out) { const auto& r = h->em.responsibilities(); std::copy(r.a.begin(), r.a.end(), out); }
void orc_em_get_labels(const OrcEM* h, unsigned* out) { const auto& l = h->em.labels(); std::copy(l.begin(), l.end(), out); }
int orc_sample_covariance(const double* x, unsigned d, unsigned n, unsigned ld, double* out)
{
    return guarded([&] {
        const Mat c = EM::calculate_sample_covariance(DataView{x, d, n, ld});
        std::copy(c.a.begin(), c.a.end(), out);
    });
}

// ---- KMeans -----------------------------------------------------------------------------------------
struct OrcKM { KMeans km; explicit OrcKM(unsigned K) : km(K) {} };

int orc_km_create(unsigned K, OrcKM** out) { return guarded([&] { *out = new OrcKM(K); }); }
void orc_km_destroy(OrcKM* h) { delete h; }
int orc_km_set_seed(OrcKM* h, unsigned s) { return guarded([&] { h->km.set_seed(s); }); }
int orc_km_set_absolute_tolerance(OrcKM* h, double t) { return guarded([&] { h->km.set_absolute_tolerance(t); }); }
int orc_km_set_maximum_steps(OrcKM* h, unsigned m) { return guarded([&] { h->km.set_maximum_steps(m); }); }
int orc_km_set_number_initialisations(OrcKM* h, unsigned n) { return guarded([&] { h->km.set_number_initialisations(n); }); }
int orc_km_set_centroids_initialiser(OrcKM* h, int kind, const double* fixed, unsigned d)
{
    return guarded([&] { h->km.set_centroids_initialiser(make_init(kind, fixed, d, h->km.K())); });
}
int orc_km_fit(OrcKM* h, const double* x, unsigned d, unsigned n, unsigned ld, int* converged)
{
    return guarded([&] { *converged = h->km.fit(DataView{x, d, n, ld}) ? 1 : 0; });
}
int orc_km_set_centroids(OrcKM* h, const double* c, unsigned d, unsigned n)
{
    return guarded([&] {
        Mat m(d, h->km.K());
        std::copy_n(c, static_cast<std::size_t>(d) * h->km.K(), m.a.data());
        h->km.set_centroids(m, n);
    });
}
int orc_km_assignment_step(OrcKM* h, const double* x, unsigned d, unsigned n, unsigned ld)
{
    return guarded([&] { h->km.assignment_step(DataView{x, d, n, ld}); });
}
int orc_km_update_step(OrcKM* h, const double* x, unsigned d, unsigned n, unsigned ld)
{
    return guarded([&] { h->km.update_step(DataView{x, d, n, ld}); });
}
int orc_km_assign_label(const OrcKM* h, const double* x, unsigned* label, double* dist2)
{
    return guarded([&] {
        const auto r = h->km.assign_label(x);
        *label = r.first;
        *dist2 = r.second;
    });
}
double orc_km_inertia(const OrcKM* h) { return h->km.inertia(); }
int orc_km_converged(const OrcKM* h) { return h->km.converged() ? 1 : 0; }
unsigned orc_km_steps_done(const OrcKM* h) { return h->km.steps_done(); }
void orc_km_get_centroids(const OrcKM* h, double* out) { const auto& c = h->km.centroids(); std::copy(c.a.begin(), c.a.end(), out); }
void orc_km_get_labels(const OrcKM* h, unsigned* out) { const auto& l = h->km.labels(); std::copy(l.begin(), l.end(), out); }

// ---- the reference tests' data sets, regenerated with the same libstdc++ <random> calls ---------------
// Two 3-d Gaussians, N samples: data layout and draw order of Tests/test_EM.cpp:10-34 and
// Tests/test_KMeans.cpp:10-36 (default-seeded engine; u01 picks the component, then one normal per dim).
// `means`/`sigmas` are 3 x 2 column-major (the numeric fixture lives with the caller).
void orc_testdata_two_gaussians(unsigned n, double p0, const double* means, const double* sigmas,
                                double* data /*3 x n*/, unsigned* truth /*n*/)
{
    std::default_random_engine rng;
    std::uniform_real_distribution<double> u01(0, 1);
    std::normal_distribution<double> standard_normal;
    for (unsigned i = 0; i < n; ++i) {
        const unsigned k = u01(rng) < p0 ? 0 : 1;
        truth[i] = k;
        for (unsigned l = 0; l < 3; ++l) data[3 * i + l] = standard_normal(rng) * sigmas[3 * k + l] + means[3 * k + l];
    }
}
// "Mousie" 2-d set of Benchmarks/bm_EM.cpp:11-36 (face + two ears, default-seeded engine).
void orc_testdata_mousie(unsigned n, double* data /*2 x n*/, unsigned* classes /*n*/)
{
    const double pi = 3.14159265358979323846;
    std::default_random_engine rng;
    std::uniform_real_distribution<double> u01(0, 1);
    const double face_radius = 1, ear_radius = 0.3;
    const double radii[3] = {face_radius, ear_radius, ear_radius};
    std::discrete_distribution<unsigned int> component{face_radius * face_radius, 2 * ear_radius * ear_radius, 2 * ear_radius * ear_radius};
    const double ear_angle = 45 * pi / 180.;
    const double cx[3] = {0, -(face_radius + ear_radius) * std::sin(ear_angle), (face_radius + ear_radius) * std::sin(ear_angle)};
    const double cy[3] = {0, (face_radius + ear_radius) * std::cos(ear_angle), (face_radius + ear_radius) * std::cos(ear_angle)};
    for (unsigned i = 0; i < n; ++i) {
        const unsigned k = component(rng);
        classes[i] = k;
        const double phi = 2 * pi * u01(rng);
        const double r = std::sqrt(u01(rng)) * radii[k];
        data[2 * i] = cx[k] + r * std::cos(phi);
        data[2 * i + 1] = cy[k] + r * std::sin(phi);
    }
}

// ---- timing helper for bench.py's cpu_baseline leg: T fixed EM iterations from given parameters ----
// Returns seconds per iteration (E-step + M-step incl. Cholesky/inverse), single thread.
double orc_em_time_iterations(OrcEM* h, const double* x, unsigned d, unsigned n, unsigned ld, unsigned iters)
{
    const DataView dv{x, d, n, ld};
    h->em.prepare_for_steps(dv);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned t = 0; t < iters; ++t) {
        h->em.expectation_step(dv);
        h->em.maximisation_step(dv);
    }
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / iters;
}
double orc_km_time_steps(OrcKM* h, const double* x, unsigned d, unsigned n, unsigned ld, unsigned iters)
{
    const DataView dv{x, d, n, ld};
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned t = 0; t < iters; ++t) {
        h->km.assignment_step(dv);
        h->km.update_step(dv);
    }
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / iters;
}

}  // extern "C"
