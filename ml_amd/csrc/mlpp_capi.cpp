// Flat C handles over the C++ facade (include/mlpp_c.h). Error text goes through the same thread-local slot as
// mlhip_last_error(): a failing facade call rethrows the C-ABI message, a facade-level exception is recorded here.
#include "mlpp_c.h"

#include <algorithm>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>

#include "host/facade/DeviceInit.hpp"
#include "ML/Clustering.hpp"
#include "ML/Device.hpp"
#include "ML/EM.hpp"
#include "ML/KMeans.hpp"
#include "ML/LinearAlgebra.hpp"
#include "ML/LinearRegression.hpp"
#include "mlhip.h"

extern "C" void mlhip_set_last_error_(const char* msg);   // defined in runtime/context.cpp

using namespace ml;

struct mlpp_centroids_initialiser { std::shared_ptr<const Clustering::CentroidsInitialiser> p; };
struct mlpp_responsibilities_initialiser { std::shared_ptr<const Clustering::ResponsibilitiesInitialiser> p; };
struct mlpp_em { EM em; Index d = 0; Index n = 0; explicit mlpp_em(unsigned K) : em(K) {} };
struct mlpp_kmeans { Clustering::KMeans km; Index d = 0; Index n = 0; explicit mlpp_kmeans(unsigned K) : km(K) {} };

namespace {
template <class F> int guarded(F&& f)
{
    try { f(); return MLHIP_OK; }
    catch (const std::invalid_argument& e) { mlhip_set_last_error_(e.what()); return MLHIP_E_INVALID_ARGUMENT; }
    catch (const std::domain_error& e) { mlhip_set_last_error_(e.what()); return MLHIP_E_DOMAIN; }
    catch (const std::exception& e) { mlhip_set_last_error_(e.what()); return MLHIP_E_RUNTIME; }
}
void need(const void* p) { if (!p) throw std::invalid_argument("null handle or pointer"); }
std::default_random_engine make_engine(int seed_set, uint32_t seed)
{
    std::default_random_engine e;
    if (seed_set) e.seed(seed);
    return e;
}
}  // namespace

extern "C" {

int mlpp_forgy_create(mlpp_centroids_initialiser** out) { return guarded([&] { need(out); *out = new mlpp_centroids_initialiser{std::make_shared<Clustering::Forgy>()}; }); }
int mlpp_random_partition_create(mlpp_centroids_initialiser** out) { return guarded([&] { need(out); *out = new mlpp_centroids_initialiser{std::make_shared<Clustering::RandomPartition>()}; }); }
int mlpp_kpp_create(mlpp_centroids_initialiser** out) { return guarded([&] { need(out); *out = new mlpp_centroids_initialiser{std::make_shared<Clustering::KPP>()}; }); }
int mlpp_fixed_centroids_create(const double* centroids, uint32_t K, uint32_t d, mlpp_centroids_initialiser** out)
{
    return guarded([&] {
        need(out); need(centroids);
        MatrixXd c(d, K);
        std::copy_n(centroids, static_cast<std::size_t>(K) * d, c.data());   // K x d row-major == d x K column-major
        *out = new mlpp_centroids_initialiser{std::make_shared<Clustering::FixedCentroids>(c)};
    });
}
int mlpp_centroids_initialiser_destroy(mlpp_centroids_initialiser* h) { delete h; return MLHIP_OK; }
int mlpp_centroids_initialiser_run(const mlpp_centroids_initialiser* h, const double* data, uint64_t n, uint32_t d, uint32_t K,
                                   int seed_set, uint32_t seed, double* centroids_out)
{
    return guarded([&] {
        need(h); need(data); need(centroids_out);
        auto prng = make_engine(seed_set, seed);
        h->p->init(ConstMatrixRef(data, d, static_cast<Index>(n)), prng, K, MatrixRef(centroids_out, d, K, d));
    });
}
int mlpp_centroids_initialiser_run_on_device(const mlpp_centroids_initialiser* h, const double* data, uint64_t n, uint32_t d, uint32_t K,
                                             int seed_set, uint32_t seed, double* centroids_out)
{
    return guarded([&] {
        need(h); need(data); need(centroids_out);
        auto prng = make_engine(seed_set, seed);
        mlhip_ctx* ctx = device::context();
        mlhip_data* dev = nullptr;
        device::check(mlhip_data_upload(ctx, data, d, n, d, &dev));
        struct Free { mlhip_data* p; ~Free() { mlhip_data_free(p); } } free_it{dev};
        Clustering::detail::init_centroids(*h->p, ConstMatrixRef(data, d, static_cast<Index>(n)), prng, K, MatrixRef(centroids_out, d, K, d), ctx, dev);
    });
}
int mlpp_closest_centroid_create(const mlpp_centroids_initialiser* ci, mlpp_responsibilities_initialiser** out)
{
    return guarded([&] {
        need(out);
        *out = new mlpp_responsibilities_initialiser{std::make_shared<Clustering::ClosestCentroid>(ci ? ci->p : nullptr)};
    });
}
int mlpp_responsibilities_initialiser_destroy(mlpp_responsibilities_initialiser* h) { delete h; return MLHIP_OK; }
int mlpp_responsibilities_initialiser_run(const mlpp_responsibilities_initialiser* h, const double* data, uint64_t n, uint32_t d,
                                          uint32_t K, int seed_set, uint32_t seed, double* resp_out)
{
    return guarded([&] {
        need(h); need(data); need(resp_out);
        auto prng = make_engine(seed_set, seed);
        h->p->init(ConstMatrixRef(data, d, static_cast<Index>(n)), prng, K, MatrixRef(resp_out, static_cast<Index>(n), K, static_cast<Index>(n)));
    });
}

// ---- EM ----
int mlpp_em_create(uint32_t K, mlpp_em** out) { return guarded([&] { need(out); *out = new mlpp_em(K); }); }
int mlpp_em_destroy(mlpp_em* h) { delete h; return MLHIP_OK; }
int mlpp_em_set_seed(mlpp_em* h, uint32_t v) { return guarded([&] { need(h); h->em.set_seed(v); }); }
int mlpp_em_set_absolute_tolerance(mlpp_em* h, double v) { return guarded([&] { need(h); h->em.set_absolute_tolerance(v); }); }
int mlpp_em_set_relative_tolerance(mlpp_em* h, double v) { return guarded([&] { need(h); h->em.set_relative_tolerance(v); }); }
int mlpp_em_set_maximum_steps(mlpp_em* h, uint32_t v) { return guarded([&] { need(h); h->em.set_maximum_steps(v); }); }
int mlpp_em_set_means_initialiser(mlpp_em* h, const mlpp_centroids_initialiser* i) { return guarded([&] { need(h); h->em.set_means_initialiser(i ? i->p : nullptr); }); }
int mlpp_em_set_responsibilities_initialiser(mlpp_em* h, const mlpp_responsibilities_initialiser* i) { return guarded([&] { need(h); h->em.set_responsibilities_initialiser(i ? i->p : nullptr); }); }
int mlpp_em_set_verbose(mlpp_em* h, int v) { return guarded([&] { need(h); h->em.set_verbose(v != 0); }); }
int mlpp_em_set_maximise_first(mlpp_em* h, int v) { return guarded([&] { need(h); h->em.set_maximise_first(v != 0); }); }
int mlpp_em_set_covariance_type(mlpp_em* h, int diagonal)
{
    return guarded([&] { need(h); h->em.set_covariance_type(diagonal ? ml::EM::CovarianceType::Diagonal : ml::EM::CovarianceType::Full); });
}
int mlpp_em_fit(mlpp_em* h, const double* data, uint64_t n, uint32_t d, int* converged)
{
    return guarded([&] {
        need(h); need(converged);
        if (n && d) need(data);
        const bool ok = h->em.fit(ConstMatrixRef(data, d, static_cast<Index>(n)));
        h->d = d; h->n = static_cast<Index>(n);
        *converged = ok ? 1 : 0;
    });
}
int mlpp_em_number_components(const mlpp_em* h, uint32_t* out) { return guarded([&] { need(h); need(out); *out = h->em.number_components(); }); }
int mlpp_em_dims(const mlpp_em* h, uint32_t* d, uint64_t* n) { return guarded([&] { need(h); if (d) *d = static_cast<uint32_t>(h->em.means().rows()); if (n) *n = static_cast<uint64_t>(h->em.labels().size()); }); }
int mlpp_em_means(const mlpp_em* h, double* out) { return guarded([&] { need(h); need(out); std::copy_n(h->em.means().data(), h->em.means().size(), out); }); }
int mlpp_em_covariance(const mlpp_em* h, uint32_t k, double* out) { return guarded([&] { need(h); need(out); const MatrixXd& c = h->em.covariance(k); std::copy_n(c.data(), c.size(), out); }); }
int mlpp_em_mixing_probabilities(const mlpp_em* h, double* out) { return guarded([&] { need(h); need(out); std::copy_n(h->em.mixing_probabilities().data(), h->em.mixing_probabilities().size(), out); }); }
int mlpp_em_responsibilities(const mlpp_em* h, double* out) { return guarded([&] { need(h); need(out); const MatrixXd& r = h->em.responsibilities(); std::copy_n(r.data(), r.size(), out); }); }
int mlpp_em_responsibilities_rows(const mlpp_em* h, uint64_t first_row, uint64_t n_rows, double* out)
{
    return guarded([&] {
        need(h); need(out);
        const MatrixXd r = h->em.responsibilities_rows(static_cast<Index>(first_row), static_cast<Index>(n_rows));
        std::copy_n(r.data(), r.size(), out);
    });
}
int mlpp_em_log_likelihood(const mlpp_em* h, double* out) { return guarded([&] { need(h); need(out); *out = h->em.log_likelihood(); }); }
int mlpp_em_labels(const mlpp_em* h, uint32_t* out) { return guarded([&] { need(h); need(out); std::copy(h->em.labels().begin(), h->em.labels().end(), out); }); }
int mlpp_em_converged(const mlpp_em* h, int* out) { return guarded([&] { need(h); need(out); *out = h->em.converged() ? 1 : 0; }); }
int mlpp_em_steps_done(const mlpp_em* h, uint32_t* out) { return guarded([&] { need(h); need(out); *out = h->em.steps_done(); }); }
int mlpp_em_assign_responsibilities(const mlpp_em* h, const double* x, uint32_t xlen, double* u, uint32_t ulen)
{
    return guarded([&] { need(h); need(x); need(u); h->em.assign_responsibilities(ConstVectorRef(x, xlen), VectorRef(u, ulen)); });
}

// ---- KMeans ----
int mlpp_kmeans_create(uint32_t K, mlpp_kmeans** out) { return guarded([&] { need(out); *out = new mlpp_kmeans(K); }); }
int mlpp_kmeans_destroy(mlpp_kmeans* h) { delete h; return MLHIP_OK; }
int mlpp_kmeans_set_seed(mlpp_kmeans* h, uint32_t v) { return guarded([&] { need(h); h->km.set_seed(v); }); }
int mlpp_kmeans_set_absolute_tolerance(mlpp_kmeans* h, double v) { return guarded([&] { need(h); h->km.set_absolute_tolerance(v); }); }
int mlpp_kmeans_set_maximum_steps(mlpp_kmeans* h, uint32_t v) { return guarded([&] { need(h); h->km.set_maximum_steps(v); }); }
int mlpp_kmeans_set_number_initialisations(mlpp_kmeans* h, uint32_t v) { return guarded([&] { need(h); h->km.set_number_initialisations(v); }); }
int mlpp_kmeans_set_centroids_initialiser(mlpp_kmeans* h, const mlpp_centroids_initialiser* i) { return guarded([&] { need(h); h->km.set_centroids_initialiser(i ? i->p : nullptr); }); }
int mlpp_kmeans_set_verbose(mlpp_kmeans* h, int v) { return guarded([&] { need(h); h->km.set_verbose(v != 0); }); }
int mlpp_kmeans_fit(mlpp_kmeans* h, const double* data, uint64_t n, uint32_t d, int* converged)
{
    return guarded([&] {
        need(h); need(converged);
        if (n && d) need(data);
        const bool ok = h->km.fit(ConstMatrixRef(data, d, static_cast<Index>(n)));
        h->d = d; h->n = static_cast<Index>(n);
        *converged = ok ? 1 : 0;
    });
}
int mlpp_kmeans_number_clusters(const mlpp_kmeans* h, uint32_t* out) { return guarded([&] { need(h); need(out); *out = h->km.number_clusters(); }); }
int mlpp_kmeans_dims(const mlpp_kmeans* h, uint32_t* d, uint64_t* n) { return guarded([&] { need(h); if (d) *d = static_cast<uint32_t>(h->km.centroids().rows()); if (n) *n = static_cast<uint64_t>(h->km.labels().size()); }); }
int mlpp_kmeans_centroids(const mlpp_kmeans* h, double* out) { return guarded([&] { need(h); need(out); std::copy_n(h->km.centroids().data(), h->km.centroids().size(), out); }); }
int mlpp_kmeans_labels(const mlpp_kmeans* h, uint32_t* out) { return guarded([&] { need(h); need(out); std::copy(h->km.labels().begin(), h->km.labels().end(), out); }); }
int mlpp_kmeans_inertia(const mlpp_kmeans* h, double* out) { return guarded([&] { need(h); need(out); *out = h->km.inertia(); }); }
int mlpp_kmeans_converged(const mlpp_kmeans* h, int* out) { return guarded([&] { need(h); need(out); *out = h->km.converged() ? 1 : 0; }); }
int mlpp_kmeans_steps_done(const mlpp_kmeans* h, uint32_t* out) { return guarded([&] { need(h); need(out); *out = h->km.steps_done(); }); }
int mlpp_kmeans_assign_label(const mlpp_kmeans* h, const double* x, uint32_t xlen, uint32_t* label, double* dist2)
{
    return guarded([&] {
        need(h); need(x); need(label); need(dist2);
        if (static_cast<Index>(xlen) != h->km.centroids().rows()) throw std::invalid_argument("Wrong x size");
        const auto r = h->km.assign_label(ConstVectorRef(x, xlen));
        *label = r.first;
        *dist2 = r.second;
    });
}

// ---- LinearAlgebra ----
int mlpp_xAx_symmetric(const double* A, uint32_t rows, uint32_t cols, const double* x, uint32_t xlen, double* out)
{
    return guarded([&] {
        need(A); need(x); need(out);
        MatrixXd m(rows, cols);
        std::copy_n(A, static_cast<std::size_t>(rows) * cols, m.data());
        *out = LinearAlgebra::xAx_symmetric(m, ConstVectorRef(x, xlen));
    });
}
int mlpp_xxT(const double* x, uint32_t n, double* dest)
{
    return guarded([&] {
        need(x); need(dest);
        VectorXd v(n);
        std::copy_n(x, n, v.data());
        MatrixXd m;
        LinearAlgebra::xxT(v, m);
        std::copy_n(m.data(), m.size(), dest);
    });
}
int mlpp_add_a_xxT(const double* x, uint32_t n, double* dest, uint32_t drows, uint32_t dcols, double a)
{
    return guarded([&] {
        need(x); need(dest);
        VectorXd v(n);
        std::copy_n(x, n, v.data());
        MatrixXd m(drows, dcols);
        std::copy_n(dest, static_cast<std::size_t>(drows) * dcols, m.data());
        LinearAlgebra::add_a_xxT(v, m, a);
        std::copy_n(m.data(), m.size(), dest);
    });
}

int mlpp_calculate_XXt_beta(const double* X, uint64_t n, uint32_t q, const double* y, uint64_t ylen, const double* lambda,
                            uint32_t lambda_len, double* XXt, double* beta)
{
    return guarded([&] {
        need(X); need(y); need(lambda); need(XXt); need(beta);
        LDLT decomposition;
        const VectorXd b = LinearRegression::calculate_XXt_beta(ConstMatrixRef(X, q, static_cast<Index>(n)),
                                                                ConstVectorRef(y, static_cast<Index>(ylen)),
                                                                MatrixRef(XXt, q, q, q), decomposition,
                                                                ConstVectorRef(lambda, lambda_len));
        std::copy_n(b.data(), b.size(), beta);
    });
}

int mlpp_device_context(mlhip_ctx** out)
{
    return guarded([&] { need(out); *out = device::context(); });
}

int mlpp_device_set_context(mlhip_ctx* ctx)
{
    return guarded([&] { device::set_context(ctx); });
}

}  // extern "C"
