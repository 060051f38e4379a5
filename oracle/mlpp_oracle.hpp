// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path. See oracle/README.md.
// Nothing under ml_amd/ or include/ may include or link this.
//
// Plain C++17, no Eigen, single thread. Column-major everywhere, one sample per column (d x N),
// exactly like the reference's Eigen::MatrixXd arguments (ML/Clustering.hpp:28-33).
#pragma once
#include <cstddef>
#include <memory>
#include <random>
#include <utility>
#include <vector>

namespace oracle {

/// Minimal column-major dense matrix.
struct Mat {
    std::size_t rows = 0, cols = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(std::size_t r, std::size_t c, double fill = 0.0) : rows(r), cols(c), a(r * c, fill) {}
    void resize(std::size_t r, std::size_t c) { rows = r; cols = c; a.resize(r * c); }
    double& operator()(std::size_t i, std::size_t j) { return a[j * rows + i]; }
    double operator()(std::size_t i, std::size_t j) const { return a[j * rows + i]; }
    double* col(std::size_t j) { return a.data() + j * rows; }
    const double* col(std::size_t j) const { return a.data() + j * rows; }
};

/// Borrowed view of d x N sample data; `ld` = distance in doubles between consecutive samples.
struct DataView {
    const double* p;
    std::size_t d, n, ld;
    const double* col(std::size_t i) const { return p + i * ld; }
};

// ---- ML/LinearAlgebra.cpp -------------------------------------------------------------------
double xAx_symmetric(const Mat& A, const double* x, std::size_t xlen);   // :8-31
void xxT(const double* x, std::size_t n, Mat& dest);                     // :33-52
void add_a_xxT(const double* x, std::size_t n, Mat& dest, double a);     // :54-73

// ---- ML/LinearRegression.cpp (the dense helper that shares the covariance contraction: SURVEY section 8 row f4) ----
/// :201-230. X is q x N (DataView), y has N entries; XXt receives X X^T + diag(lambda); returns beta.
std::vector<double> calculate_XXt_beta(const DataView& X, const double* y, std::size_t ylen, Mat& XXt,
                                       const double* lambda, std::size_t lambda_len);

// ---- ML/Clustering.cpp ----------------------------------------------------------------------
struct CentroidsInitialiser {
    virtual ~CentroidsInitialiser() = default;
    virtual void init(const DataView& data, std::default_random_engine& prng, unsigned K, Mat& centroids) const = 0;
};
struct Forgy : CentroidsInitialiser {            // :16-25
    void init(const DataView&, std::default_random_engine&, unsigned, Mat&) const override;
};
struct RandomPartition : CentroidsInitialiser {  // :27-37
    void init(const DataView&, std::default_random_engine&, unsigned, Mat&) const override;
};
struct KPP : CentroidsInitialiser {              // :39-59
    void init(const DataView&, std::default_random_engine&, unsigned, Mat&) const override;
};
/// Not in the reference: returns caller-given centroids (a user subclass, allowed by ML/Clustering.hpp:58-72).
struct FixedCentroids : CentroidsInitialiser {
    Mat c;
    void init(const DataView&, std::default_random_engine&, unsigned, Mat&) const override;
};
struct ResponsibilitiesInitialiser {
    virtual ~ResponsibilitiesInitialiser() = default;
    virtual void init(const DataView& data, std::default_random_engine& prng, unsigned K, Mat& resp) const = 0;
};
struct ClosestCentroid : ResponsibilitiesInitialiser {  // :61-89
    std::shared_ptr<const CentroidsInitialiser> ci;
    explicit ClosestCentroid(std::shared_ptr<const CentroidsInitialiser> c);
    void init(const DataView&, std::default_random_engine&, unsigned, Mat&) const override;
};

// ---- ML/EM.cpp ------------------------------------------------------------------------------
class EM {
public:
    explicit EM(unsigned number_components);                 // :17-37
    void set_seed(unsigned seed) { prng_.seed(seed); }       // :39-42
    void set_absolute_tolerance(double t);                   // :44-50
    void set_relative_tolerance(double t);                   // :52-58
    void set_maximum_steps(unsigned m);                      // :60-66
    void set_means_initialiser(std::shared_ptr<const CentroidsInitialiser> p);                  // :68-74
    void set_responsibilities_initialiser(std::shared_ptr<const ResponsibilitiesInitialiser> p); // :76-82
    void set_maximise_first(bool b) { maximise_first_ = b; }
    /// EXTENSION, not in the reference (ml::EM is full-covariance only, ML/EM.hpp:175): diagonal covariances. The same loops
    /// (E-step :190-219, M-step :221-263, process_covariances :274-287) restricted to the diagonal: q = sum_j A_jj x_j^2
    /// (the diagonal terms of xAx_symmetric, :20-22), cov_jj += a x_j x_j (the diagonal terms of add_a_xxT, :63-64),
    /// L_jj = sqrt(cov_jj), inverse_jj = (1 / L_jj) / L_jj (what llt.solve(I) yields for a diagonal factor), sqrt_det =
    /// prod L_jj. Off-diagonal entries of every covariance stay 0. BASELINE.json configs[1] asks for it; the third-party
    /// pin is scikit-learn's covariance_type='diag' (tests/golden/em_onestep_diag_*.npz).
    void set_diagonal(bool b) { diagonal_ = b; }
    bool diagonal() const { return diagonal_; }
    bool fit(const DataView& data);                          // :91-174
    void assign_responsibilities(const double* x, std::size_t xlen, double* u, std::size_t ulen) const;  // :176-188

    // Single steps from explicitly given parameters (used for per-iteration parity checks).
    void set_parameters(const Mat& means, const std::vector<Mat>& covs, const std::vector<double>& pis);
    void expectation_step(const DataView& data);             // :190-219
    void maximisation_step(const DataView& data);            // :221-263
    void calculate_labels();                                 // :289-304
    static Mat calculate_sample_covariance(const DataView& data);  // :265-272
    void prepare_for_steps(const DataView& data);            // resizes work buffers like fit() does (:103-106,:139)

    unsigned K() const { return K_; }
    const Mat& means() const { return means_; }
    const std::vector<Mat>& covariances() const { return cov_; }
    const std::vector<Mat>& inverse_covariances() const { return inv_cov_; }
    const std::vector<double>& sqrt_dets() const { return sqrt_det_; }
    const std::vector<double>& mixing_probabilities() const { return pi_; }
    const Mat& responsibilities() const { return resp_; }
    Mat& mutable_responsibilities() { return resp_; }
    double log_likelihood() const { return ll_; }
    const std::vector<unsigned>& labels() const { return labels_; }
    bool converged() const { return converged_; }
    unsigned steps_done() const { return steps_done_; }

private:
    void process_covariances(std::size_t d);                 // :274-287
    std::default_random_engine prng_;
    std::shared_ptr<const CentroidsInitialiser> means_init_;
    std::shared_ptr<const ResponsibilitiesInitialiser> resp_init_;
    std::vector<double> pi_;
    Mat means_, resp_;
    std::vector<double> work_;
    std::vector<Mat> cov_, inv_cov_, chol_;
    std::vector<double> sqrt_det_;
    std::vector<unsigned> labels_;
    double atol_ = 1e-8, rtol_ = 1e-8, ll_ = 0;
    unsigned K_, max_steps_ = 1000, steps_done_ = 0;
    bool maximise_first_ = false, converged_ = false, diagonal_ = false;
};

// ---- ML/KMeans.cpp --------------------------------------------------------------------------
class KMeans {
public:
    explicit KMeans(unsigned number_clusters);               // :10-23
    void set_seed(unsigned seed) { prng_.seed(seed); }       // :116-119
    void set_absolute_tolerance(double t);                   // :121-127
    void set_maximum_steps(unsigned m);                      // :129-135
    void set_number_initialisations(unsigned n);             // :137-143
    void set_centroids_initialiser(std::shared_ptr<const CentroidsInitialiser> p);  // :145-151
    bool fit(const DataView& data);                          // :25-48
    std::pair<unsigned, double> assign_label(const double* x) const;  // :153-165

    // Single steps from explicitly given centroids.
    void set_centroids(const Mat& c, std::size_t n);
    void assignment_step(const DataView& data);              // :167-178
    void update_step(const DataView& data);                  // :180-192

    unsigned K() const { return K_; }
    const Mat& centroids() const { return c_; }
    const std::vector<unsigned>& labels() const { return labels_; }
    double inertia() const { return inertia_; }
    bool converged() const { return converged_; }
    unsigned steps_done() const { return steps_done_; }

private:
    bool fit_once(const DataView& data);                     // :50-114
    std::vector<unsigned> labels_, old_labels_;
    Mat c_, old_c_;
    std::vector<double> work_;
    std::default_random_engine prng_;
    std::shared_ptr<const CentroidsInitialiser> init_;
    double atol_ = 1e-8, inertia_ = 0;
    unsigned max_steps_ = 1000, num_inits_ = 1, K_, steps_done_ = 0;
    bool converged_ = false;
};

}  // namespace oracle
