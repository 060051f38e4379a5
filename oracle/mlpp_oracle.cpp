// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path. See oracle/README.md.
//
// Each function follows the cited reference lines operation for operation (same loop nesting,
// same association of multiplications/divisions, same strict-inequality tie-breaks, same
// libstdc++ <random> calls). Where the reference hands an expression to Eigen (GEMM, vectorised
// reductions, selfadjointView product, LLT) the restatement uses the plain sequential loop that
// defines the same mathematical result; Eigen's internal blocking / SIMD summation order is not
// reproducible without Eigen, so those places agree with the reference to rounding (~1e-16
// relative per operation), not bit for bit. They are marked [eigen-order] below.
#include "mlpp_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <iterator>
#include <limits>
#include <numeric>
#include <stdexcept>

namespace oracle {

static constexpr double kPi = 3.14159265358979323846;  // ML/EM.cpp:14

// =================================================================================================
// ML/LinearAlgebra.cpp
// =================================================================================================

// ML/LinearAlgebra.cpp:8-31
double xAx_symmetric(const Mat& A, const double* x, std::size_t xlen)
{
    if (A.rows != A.cols) throw std::invalid_argument("A matrix is not square");
    if (xlen != A.rows) throw std::invalid_argument("x has wrong size");
    const std::size_t dim = A.rows;
    if (dim < 15) {
        // :17-27 -- upper triangle only, diagonal term first, then the doubled off-diagonals of column i.
        double sum = 0;
        for (std::size_t i = 0; i < dim; ++i) {
            const double x_i = x[i];
            sum += A(i, i) * x_i * x_i;
            for (std::size_t j = 0; j < i; ++j) sum += 2 * A(j, i) * x_i * x[j];
        }
        return sum;
    }
    // :29 -- x^T * (A.selfadjointView<Upper>() * x): symmetric matrix-vector product reading only
    // the upper triangle, then a dot product. [eigen-order]
    double result = 0;
    for (std::size_t i = 0; i < dim; ++i) {
        double t = 0;
        for (std::size_t j = 0; j < dim; ++j) t += (j >= i ? A(i, j) : A(j, i)) * x[j];
        result += x[i] * t;
    }
    return result;
}

// ML/LinearAlgebra.cpp:33-52
void xxT(const double* x, std::size_t n, Mat& dest)
{
    if (dest.rows != n || dest.cols != n) dest.resize(n, n);
    if (n < 11) {
        for (std::size_t i = 0; i < n; ++i) {
            const double x_i = x[i];
            dest(i, i) = x_i * x_i;
            for (std::size_t j = 0; j < i; ++j) {
                const double v = x_i * x[j];
                dest(i, j) = v;
                dest(j, i) = v;
            }
        }
    } else {
        for (std::size_t j = 0; j < n; ++j)   // :50 outer product x * x^T
            for (std::size_t i = 0; i < n; ++i) dest(i, j) = x[i] * x[j];
    }
}

// ML/LinearAlgebra.cpp:54-73
void add_a_xxT(const double* x, std::size_t n, Mat& dest, const double a)
{
    if (dest.rows != n || dest.cols != n) throw std::invalid_argument("Expected square matrix with the same size as x");
    if (n < 14) {
        for (std::size_t i = 0; i < n; ++i) {
            const double x_i = x[i];
            dest(i, i) += a * x_i * x_i;
            for (std::size_t j = 0; j < i; ++j) {
                const double v = a * x_i * x[j];
                dest(i, j) += v;
                dest(j, i) += v;
            }
        }
    } else {
        // :71 dest += (a * x) * x^T : Eigen materialises a*x, then column j += x[j] * (a*x). [eigen-order]
        for (std::size_t j = 0; j < n; ++j)
            for (std::size_t i = 0; i < n; ++i) dest(i, j) += x[j] * (a * x[i]);
    }
}

// =================================================================================================
// ML/LinearRegression.cpp:201-230
// =================================================================================================
std::vector<double> calculate_XXt_beta(const DataView& X, const double* y, std::size_t ylen, Mat& XXt,
                                       const double* lambda, std::size_t lambda_len)
{
    const std::size_t n = X.n, q = X.d;
    double min_lambda = lambda_len ? lambda[0] : 0.0;
    for (std::size_t i = 1; i < lambda_len; ++i) min_lambda = std::min(min_lambda, lambda[i]);
    if (min_lambda < 0) throw std::domain_error("Ridge regularisation constant cannot be negative");          // :206-208
    if (lambda_len != q) throw std::invalid_argument("Lambda vector must have the same size as the number of features");
    if (n != ylen) throw std::invalid_argument("X matrix has different number of data points than Y has values");
    if (n < q) throw std::invalid_argument("Not enough data points for regression");
    std::vector<double> b(q, 0.0);                                      // :218  b = X * y  [eigen-order]
    for (std::size_t i = 0; i < n; ++i)
        for (std::size_t a = 0; a < q; ++a) b[a] += X.col(i)[a] * y[i];
    XXt = Mat(q, q, 0.0);                                               // :220  XXt = X * X^T  [eigen-order]
    for (std::size_t i = 0; i < n; ++i) {
        const double* x = X.col(i);
        for (std::size_t c = 0; c < q; ++c)
            for (std::size_t a = 0; a < q; ++a) XXt(a, c) += x[a] * x[c];
    }
    if (min_lambda != 0)                                                // :221-225 `if (lambda.minCoeff())`: the ridge is
        for (std::size_t i = 0; i < q; ++i) XXt(i, i) += lambda[i];     //   skipped altogether when its smallest entry is 0
    // :228-229 xxt_decomp.compute(XXt); solve(b)  -- LDL^T; unpivoted here (the matrices used are positive definite)
    Mat L(q, q, 0.0);
    std::vector<double> D(q);
    for (std::size_t j = 0; j < q; ++j) {
        double dj = XXt(j, j);
        for (std::size_t l = 0; l < j; ++l) dj -= L(j, l) * L(j, l) * D[l];
        D[j] = dj;
        L(j, j) = 1;
        for (std::size_t i = j + 1; i < q; ++i) {
            double t = XXt(i, j);
            for (std::size_t l = 0; l < j; ++l) t -= L(i, l) * L(j, l) * D[l];
            L(i, j) = t / dj;
        }
    }
    std::vector<double> beta(b);
    for (std::size_t i = 0; i < q; ++i)
        for (std::size_t l = 0; l < i; ++l) beta[i] -= L(i, l) * beta[l];
    for (std::size_t i = 0; i < q; ++i) beta[i] /= D[i];
    for (std::size_t i = q; i-- > 0;)
        for (std::size_t l = i + 1; l < q; ++l) beta[i] -= L(l, i) * beta[l];
    return beta;
}

// =================================================================================================
// ML/Clustering.cpp
// =================================================================================================

// (x - c).squaredNorm() as used at ML/Clustering.cpp:47,78,81 and ML/KMeans.cpp:158. [eigen-order]
// The reference's release build (-march=native, SConstruct:20) fuses the multiply-add; std::fma pins that choice
// here independently of the compiler's contraction setting.
static double squared_distance(const double* x, const double* c, std::size_t d)
{
    double s = 0;
    for (std::size_t j = 0; j < d; ++j) {
        const double t = x[j] - c[j];
        s = std::fma(t, t, s);
    }
    return s;
}

// ML/Clustering.cpp:16-25
void Forgy::init(const DataView& data, std::default_random_engine& prng, unsigned K, Mat& centroids) const
{
    std::vector<std::ptrdiff_t> all_indices(data.n);   // Eigen::Index == std::ptrdiff_t
    std::iota(all_indices.begin(), all_indices.end(), 0);
    std::vector<std::ptrdiff_t> sampled;
    std::sample(all_indices.begin(), all_indices.end(), std::back_inserter(sampled), K, prng);
    for (unsigned i = 0; i < K; ++i)
        std::copy_n(data.col(static_cast<std::size_t>(sampled[i])), data.d, centroids.col(i));
}

// ML/Clustering.cpp:27-37
void RandomPartition::init(const DataView& data, std::default_random_engine& prng, unsigned K, Mat& centroids) const
{
    std::fill(centroids.a.begin(), centroids.a.end(), 0.0);
    std::vector<unsigned> counters(K, 0);
    std::uniform_int_distribution<unsigned int> dist(0, K - 1);
    for (std::size_t i = 0; i < data.n; ++i) {
        const unsigned k = dist(prng);
        const double denom = static_cast<double>(++counters[k]);
        double* c = centroids.col(k);
        const double* x = data.col(i);
        for (std::size_t j = 0; j < data.d; ++j) c[j] += (x[j] - c[j]) / denom;
    }
}

// ML/Clustering.cpp:39-59
void KPP::init(const DataView& data, std::default_random_engine& prng, unsigned K, Mat& centroids) const
{
    std::vector<double> weights(data.n);
    for (unsigned n = 0; n < K; ++n) {
        if (n) {
            for (std::size_t i = 0; i < data.n; ++i) {
                double best = std::numeric_limits<double>::infinity();
                for (unsigned k = 0; k < n; ++k)
                    best = std::min(best, squared_distance(data.col(i), centroids.col(k), data.d));
                weights[i] = best;
            }
        } else {
            std::fill(weights.begin(), weights.end(), 1);
        }
        std::discrete_distribution<std::ptrdiff_t> dist(weights.begin(), weights.end());
        const auto idx = dist(prng);
        std::copy_n(data.col(static_cast<std::size_t>(idx)), data.d, centroids.col(n));
    }
}

void FixedCentroids::init(const DataView& data, std::default_random_engine&, unsigned K, Mat& centroids) const
{
    if (c.rows != data.d || c.cols != K) throw std::invalid_argument("FixedCentroids: shape mismatch");
    centroids = c;
}

// ML/Clustering.cpp:64-70
ClosestCentroid::ClosestCentroid(std::shared_ptr<const CentroidsInitialiser> c) : ci(std::move(c))
{
    if (!ci) throw std::invalid_argument("Null centroids initialiser");
}

// ML/Clustering.cpp:72-89
void ClosestCentroid::init(const DataView& data, std::default_random_engine& prng, unsigned K, Mat& resp) const
{
    Mat centroids(data.d, K);
    ci->init(data, prng, K, centroids);
    std::fill(resp.a.begin(), resp.a.end(), 0.0);
    for (std::size_t i = 0; i < data.n; ++i) {
        double best = squared_distance(data.col(i), centroids.col(0), data.d);
        unsigned arg = 0;
        for (unsigned k = 1; k < K; ++k) {
            const double dist = squared_distance(data.col(i), centroids.col(k), data.d);
            if (dist < best) { best = dist; arg = k; }
        }
        resp(i, arg) = 1;
    }
}

// =================================================================================================
// ML/EM.cpp
// =================================================================================================

// ML/EM.cpp:17-37 (defaults :19-32)
EM::EM(unsigned number_components)
    : means_init_(std::make_shared<Forgy>())
    , resp_init_(std::make_shared<ClosestCentroid>(means_init_))
    , pi_(number_components)
    , cov_(number_components), inv_cov_(number_components), chol_(number_components)
    , sqrt_det_(number_components)
    , K_(number_components)
{
    if (!number_components) throw std::invalid_argument("EM: At least one component required");
}

void EM::set_absolute_tolerance(double t) { if (t < 0) throw std::domain_error("EM: Negative absolute tolerance"); atol_ = t; }
void EM::set_relative_tolerance(double t) { if (t < 0) throw std::domain_error("EM: Negative relative tolerance"); rtol_ = t; }
void EM::set_maximum_steps(unsigned m) { if (m < 2) throw std::invalid_argument("EM: At least two steps required for convergence test"); max_steps_ = m; }
void EM::set_means_initialiser(std::shared_ptr<const CentroidsInitialiser> p) { if (!p) throw std::invalid_argument("EM: Null means initialiser"); means_init_ = std::move(p); }
void EM::set_responsibilities_initialiser(std::shared_ptr<const ResponsibilitiesInitialiser> p) { if (!p) throw std::invalid_argument("EM: Null responsibilities initialiser"); resp_init_ = std::move(p); }

void EM::prepare_for_steps(const DataView& data)
{
    means_.resize(data.d, K_);          // :103
    resp_.resize(data.n, K_);           // :104
    labels_.resize(data.n);             // :106
    work_.resize(data.d);               // :139
    for (unsigned k = 0; k < K_; ++k) if (cov_[k].rows != data.d) cov_[k] = Mat(data.d, data.d);
}

void EM::set_parameters(const Mat& means, const std::vector<Mat>& covs, const std::vector<double>& pis)
{
    means_ = means; cov_ = covs; pi_ = pis;
    process_covariances(means.rows);
}

// ML/EM.cpp:91-174
bool EM::fit(const DataView& data)
{
    converged_ = false;
    steps_done_ = 0;
    const std::size_t d = data.d, n = data.n;
    if (!d) throw std::invalid_argument("EM: At least one dimension required");
    if (n < K_) throw std::invalid_argument("EM: Not enough data ");

    means_.resize(d, K_);
    resp_.resize(n, K_);
    std::fill(pi_.begin(), pi_.end(), 1. / static_cast<double>(K_));
    labels_.resize(n);

    if (n == K_) {
        // :108-118 exact deterministic fit.
        std::fill(resp_.a.begin(), resp_.a.end(), 0.0);
        for (unsigned i = 0; i < K_; ++i) {
            resp_(i, i) = 1;
            std::copy_n(data.col(i), d, means_.col(i));
            cov_[i] = Mat(d, d, 0.0);
            ll_ = std::numeric_limits<double>::infinity();
            labels_[i] = i;
        }
        converged_ = true;
    } else {
        if (maximise_first_) {
            resp_init_->init(data, prng_, K_, resp_);                 // :121
            for (unsigned k = 0; k < K_; ++k) cov_[k].resize(d, d);   // :122-124 (contents unspecified; *= 0 follows)
            for (unsigned k = 0; k < K_; ++k) std::fill(cov_[k].a.begin(), cov_[k].a.end(), 0.0);
            maximisation_step(data);                                  // :125
        } else {
            means_init_->init(data, prng_, K_, means_);               // :128
            Mat sample_cov = calculate_sample_covariance(data);       // :129
            if (diagonal_)                                            // extension: keep the variances only
                for (std::size_t b = 0; b < d; ++b)
                    for (std::size_t a = 0; a < d; ++a)
                        if (a != b) sample_cov(a, b) = 0.0;
            for (unsigned k = 0; k < K_; ++k) cov_[k] = sample_cov;   // :132-134
            process_covariances(d);                                   // :135
        }
        work_.resize(d);
        double old_ll = -std::numeric_limits<double>::infinity();
        for (unsigned step = 0; step < max_steps_; ++step) {          // :143
            expectation_step(data);
            maximisation_step(data);
            ++steps_done_;
            if (step > 0) {                                           // :161-168
                const double ll_change = std::abs(ll_ - old_ll);
                if (ll_change < atol_ + rtol_ * std::max(std::abs(old_ll), std::abs(ll_))) {
                    calculate_labels();
                    converged_ = true;
                    break;
                }
            }
            old_ll = ll_;
        }
    }
    return converged_;
}

// ML/EM.cpp:176-188
void EM::assign_responsibilities(const double* x, std::size_t xlen, double* u, std::size_t ulen) const
{
    if (xlen != means_.rows) throw std::invalid_argument("Wrong x size");
    if (ulen != K_) throw std::invalid_argument("Wrong u size");
    std::vector<double> diff(xlen);
    for (unsigned k = 0; k < K_; ++k) {
        for (std::size_t j = 0; j < xlen; ++j) diff[j] = x[j] - means_(j, k);
        // :185 association: (exp(..) * pi_k) / sqrt_det_k
        u[k] = std::exp(-0.5 * xAx_symmetric(inv_cov_[k], diff.data(), xlen)) * pi_[k] / sqrt_det_[k];
    }
    double s = 0;
    for (unsigned k = 0; k < K_; ++k) s += u[k];
    for (unsigned k = 0; k < K_; ++k) u[k] /= s;
}

// ML/EM.cpp:190-219
void EM::expectation_step(const DataView& data)
{
    const std::size_t d = data.d, n = data.n;
    work_.resize(d);   // Eigen's `work_vector_ = ...` assignment resizes (:206)
    static const double log_2_pi = std::log(2. * kPi);
    const double ll_norm = static_cast<double>(d) * log_2_pi / 2;

    for (unsigned k = 0; k < K_; ++k) {                               // :201
        const double* mean = means_.col(k);
        double* w = resp_.col(k);
        const Mat& inv = inv_cov_[k];
        for (std::size_t i = 0; i < n; ++i) {                         // :205-208
            const double* x = data.col(i);
            for (std::size_t j = 0; j < d; ++j) work_[j] = x[j] - mean[j];
            if (diagonal_) {
                double q = 0;                                         // the diagonal terms of xAx_symmetric (:20-22)
                for (std::size_t j = 0; j < d; ++j) q += inv(j, j) * work_[j] * work_[j];
                w[i] = std::exp(-0.5 * q);
            } else {
                w[i] = std::exp(-0.5 * xAx_symmetric(inv, work_.data(), d));
            }
        }
        const double scale = pi_[k] / sqrt_det_[k];                   // :209  column *= (pi/sqrt_det)
        for (std::size_t i = 0; i < n; ++i) w[i] *= scale;
    }
    // :211 rowwise().sum().array().log().mean() - const. [eigen-order] for the outer mean.
    double sum_logs = 0;
    for (std::size_t i = 0; i < n; ++i) {
        double s = 0;
        for (unsigned k = 0; k < K_; ++k) s += resp_(i, k);
        sum_logs += std::log(s);
    }
    ll_ = sum_logs / static_cast<double>(n) - ll_norm;

    for (std::size_t i = 0; i < n; ++i) {                             // :214-218
        double s = 0;
        for (unsigned k = 0; k < K_; ++k) s += resp_(i, k);
        for (unsigned k = 0; k < K_; ++k) resp_(i, k) /= s;
    }
}

// ML/EM.cpp:221-263
void EM::maximisation_step(const DataView& data)
{
    const std::size_t d = data.d, n = data.n;
    work_.resize(d);   // Eigen's `work_vector_ = ...` assignment resizes (:246); fit() calls this before :139
    // :229 means = data * responsibilities (unnormalised). [eigen-order]
    for (unsigned k = 0; k < K_; ++k) {
        double* m = means_.col(k);
        std::fill(m, m + d, 0.0);
        const double* w = resp_.col(k);
        for (std::size_t i = 0; i < n; ++i) {
            const double* x = data.col(i);
            for (std::size_t j = 0; j < d; ++j) m[j] += x[j] * w[i];
        }
    }
    for (unsigned k = 0; k < K_; ++k) {                               // :234
        Mat& cov = cov_[k];
        for (double& v : cov.a) v *= 0;                               // :236 (NaN-preserving)
        const double* w = resp_.col(k);
        double sum_w = 0;                                             // :238 [eigen-order]
        for (std::size_t i = 0; i < n; ++i) sum_w += w[i];
        double* mean = means_.col(k);
        for (std::size_t j = 0; j < d; ++j) mean[j] /= sum_w;         // :242
        for (std::size_t i = 0; i < n; ++i) {                         // :245-248
            const double* x = data.col(i);
            for (std::size_t j = 0; j < d; ++j) work_[j] = x[j] - mean[j];
            if (diagonal_) {
                for (std::size_t j = 0; j < d; ++j) cov(j, j) += w[i] * work_[j] * work_[j];   // add_a_xxT's diagonal (:63-64)
            } else {
                add_a_xxT(work_.data(), d, cov, w[i]);
            }
        }
        for (double& v : cov.a) v /= sum_w;                           // :250
        static constexpr double epsilon = 1e-15;                      // :252
        for (std::size_t j = 0; j < d; ++j) cov(j, j) += epsilon;
        pi_[k] = sum_w / static_cast<double>(n);                      // :257
    }
    process_covariances(d);                                           // :262
}

// ML/EM.cpp:265-272
Mat EM::calculate_sample_covariance(const DataView& data)
{
    const std::size_t d = data.d, n = data.n;
    std::vector<double> mean(d, 0.0);
    for (std::size_t i = 0; i < n; ++i)
        for (std::size_t j = 0; j < d; ++j) mean[j] += data.col(i)[j];
    for (std::size_t j = 0; j < d; ++j) mean[j] /= static_cast<double>(n);
    Mat cov(d, d, 0.0);
    std::vector<double> c(d);
    for (std::size_t i = 0; i < n; ++i) {                             // centred * centred^T [eigen-order]
        for (std::size_t j = 0; j < d; ++j) c[j] = data.col(i)[j] - mean[j];
        for (std::size_t b = 0; b < d; ++b)
            for (std::size_t a = 0; a < d; ++a) cov(a, b) += c[a] * c[b];
    }
    for (double& v : cov.a) v /= static_cast<double>(n - 1);
    return cov;
}

// Lower Cholesky factor of a symmetric matrix (what Eigen::LLT<MatrixXd>::compute produces, :279). [eigen-order]
static void cholesky_lower(const Mat& A, Mat& L)
{
    const std::size_t d = A.rows;
    L = Mat(d, d, 0.0);
    for (std::size_t j = 0; j < d; ++j) {
        double s = A(j, j);
        for (std::size_t l = 0; l < j; ++l) s -= L(j, l) * L(j, l);
        const double ljj = std::sqrt(s);
        L(j, j) = ljj;
        for (std::size_t i = j + 1; i < d; ++i) {
            double t = A(i, j);
            for (std::size_t l = 0; l < j; ++l) t -= L(i, l) * L(j, l);
            L(i, j) = t / ljj;
        }
    }
}

// ML/EM.cpp:274-287
void EM::process_covariances(std::size_t d)
{
    for (unsigned k = 0; k < K_ && diagonal_; ++k) {
        // extension: the same decomposition of a diagonal matrix, without the O(d^3) loops
        Mat L(d, d, 0.0), inv(d, d, 0.0);
        double sd = 1;
        for (std::size_t j = 0; j < d; ++j) {
            const double ljj = std::sqrt(cov_[k](j, j));
            L(j, j) = ljj;
            inv(j, j) = (1.0 / ljj) / ljj;
            sd *= ljj;
        }
        chol_[k] = L;
        inv_cov_[k] = inv;
        sqrt_det_[k] = sd;
    }
    if (diagonal_) return;
    for (unsigned k = 0; k < K_; ++k) {
        cholesky_lower(cov_[k], chol_[k]);                            // :279
        const Mat& L = chol_[k];
        // :280 inverse = llt.solve(Identity): L y = e_c, then L^T x = y, column by column.
        Mat inv(d, d, 0.0);
        std::vector<double> y(d);
        for (std::size_t c = 0; c < d; ++c) {
            for (std::size_t i = 0; i < d; ++i) {
                double t = (i == c) ? 1.0 : 0.0;
                for (std::size_t l = 0; l < i; ++l) t -= L(i, l) * y[l];
                y[i] = t / L(i, i);
            }
            for (std::size_t ii = d; ii-- > 0;) {
                double t = y[ii];
                for (std::size_t l = ii + 1; l < d; ++l) t -= L(l, ii) * inv(l, c);
                inv(ii, c) = t / L(ii, ii);
            }
        }
        inv_cov_[k] = inv;
        double sd = 1;                                                // :281-284
        for (std::size_t i = 0; i < d; ++i) sd *= L(i, i);
        sqrt_det_[k] = sd;
    }
}

// ML/EM.cpp:289-304
void EM::calculate_labels()
{
    for (std::size_t i = 0; i < resp_.rows; ++i) {
        double best = -1;
        long label = -1;
        for (std::size_t k = 0; k < resp_.cols; ++k) {
            if (resp_(i, k) > best) { best = resp_(i, k); label = static_cast<long>(k); }
        }
        labels_[i] = static_cast<unsigned>(label);
    }
}

// =================================================================================================
// ML/KMeans.cpp
// =================================================================================================

// ML/KMeans.cpp:10-23
KMeans::KMeans(unsigned number_clusters)
    : work_(number_clusters), init_(std::make_shared<Forgy>()), K_(number_clusters)
{
    if (!number_clusters) throw std::invalid_argument("KMeans: number of clusters cannot be zero");
}

void KMeans::set_absolute_tolerance(double t) { if (t < 0) throw std::domain_error("KMeans: Negative absolute tolerance"); atol_ = t; }
void KMeans::set_maximum_steps(unsigned m) { if (m < 2) throw std::invalid_argument("KMeans: At least two steps required for convergence test"); max_steps_ = m; }
void KMeans::set_number_initialisations(unsigned n) { if (n < 1) throw std::invalid_argument("KMeans: At least 1 initialisation required"); num_inits_ = n; }
void KMeans::set_centroids_initialiser(std::shared_ptr<const CentroidsInitialiser> p) { if (!p) throw std::invalid_argument("KMeans: Null centroids initialiser"); init_ = std::move(p); }

void KMeans::set_centroids(const Mat& c, std::size_t n)
{
    c_ = c;
    old_c_.resize(c.rows, c.cols);
    labels_.resize(n);
    old_labels_.resize(n);
}

// ML/KMeans.cpp:25-48
bool KMeans::fit(const DataView& data)
{
    if (num_inits_ == 1) return fit_once(data);
    converged_ = false;
    double min_inertia = std::numeric_limits<double>::infinity();
    Mat best;
    for (unsigned i = 0; i < num_inits_; ++i) {
        if (fit_once(data)) {
            if (inertia_ < min_inertia) { min_inertia = inertia_; best = c_; }
            converged_ = true;
        }
    }
    if (converged_) {
        c_ = best;
        assignment_step(data);
    }
    return converged_;
}

// ML/KMeans.cpp:50-114
bool KMeans::fit_once(const DataView& data)
{
    converged_ = false;
    steps_done_ = 0;
    const std::size_t d = data.d, n = data.n;
    if (!d) throw std::invalid_argument("KMeans: At least one dimension required");
    if (n < K_) throw std::invalid_argument("KMeans: Not enough data ");
    c_.resize(d, K_);
    old_c_.resize(d, K_);
    labels_.resize(n);
    old_labels_.resize(n);
    if (n == K_) {
        for (unsigned i = 0; i < K_; ++i) {
            std::copy_n(data.col(i), d, c_.col(i));
            labels_[i] = i;
        }
        inertia_ = 0;
        converged_ = true;
    } else {
        init_->init(data, prng_, K_, c_);
        for (unsigned step = 0; step < max_steps_; ++step) {
            assignment_step(data);
            ++steps_done_;
            if (step > 0 && old_labels_ == labels_) { converged_ = true; break; }
            update_step(data);
            if (step > 0) {
                double shift = 0;                                     // (c - old).squaredNorm() :103 [eigen-order]
                for (std::size_t t = 0; t < c_.a.size(); ++t) {
                    const double dlt = c_.a[t] - old_c_.a[t];
                    shift += dlt * dlt;
                }
                if (shift < atol_) {
                    assignment_step(data);
                    converged_ = true;
                    break;
                }
            }
        }
    }
    return converged_;
}

// ML/KMeans.cpp:153-165
std::pair<unsigned, double> KMeans::assign_label(const double* x) const
{
    double best = std::numeric_limits<double>::infinity();
    unsigned label = 0;
    for (unsigned k = 0; k < K_; ++k) {
        const double dist = squared_distance(x, c_.col(k), c_.rows);
        if (dist < best) { best = dist; label = k; }
    }
    return {label, best};
}

// ML/KMeans.cpp:167-178
void KMeans::assignment_step(const DataView& data)
{
    old_labels_.swap(labels_);
    inertia_ = 0;
    for (std::size_t i = 0; i < data.n; ++i) {
        const auto ld = assign_label(data.col(i));
        labels_[i] = ld.first;
        inertia_ += ld.second;
    }
}

// ML/KMeans.cpp:180-192
void KMeans::update_step(const DataView& data)
{
    std::fill(work_.begin(), work_.end(), 0.0);
    std::swap(old_c_, c_);
    c_.resize(data.d, K_);
    std::fill(c_.a.begin(), c_.a.end(), 0.0);
    for (std::size_t i = 0; i < data.n; ++i) {
        const unsigned label = labels_[i];
        const double num = (++work_[label]);
        double* c = c_.col(label);
        const double* x = data.col(i);
        for (std::size_t j = 0; j < data.d; ++j) c[j] += (x[j] - c[j]) / num;
    }
}

}  // namespace oracle
