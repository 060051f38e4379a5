// Host-side dense helpers of the EM path (tiny d x d work that stays on the CPU):
// what EM::process_covariances (reference ML/EM.cpp:274-287) and the closing lines of
// EM::maximisation_step (ML/EM.cpp:242, 250-257) do, expressed on raw column-major buffers.
#pragma once
#include <cstddef>
#include <cstdint>

namespace mlhip {
namespace host {

/// Lower Cholesky factor of the symmetric d x d matrix A (column-major); what Eigen::LLT computes at
/// ML/EM.cpp:279. A non-positive-definite A yields NaNs, silently, like the reference (no info() check).
void cholesky_lower(int d, const double* A, double* L);

/// inverse = A^-1 via L y = e_c, L^T x = y (the llt.solve(Identity) of ML/EM.cpp:280);
/// sqrt_det = prod L_ii (ML/EM.cpp:281-285).
void process_covariance(int d, const double* cov, double* inverse, double* sqrt_det);

/// One E-step record per component for the device kernel (layout: device/device.hpp estep_param_stride):
/// [mean(D) | W = L^-1 packed lower, row-major | log(pi) - sum log L_jj], coordinates >= d zero-padded.
void build_estep_params(int d, int D, int K, const double* mixing, const double* means, const double* covariances,
                        double* records);

/// Same for the matrix-core E-step kernel (layout: device/layout.hpp estep_mfma_param_stride).
void build_estep_params_mfma(int d, int D, int K, const double* mixing, const double* means, const double* covariances,
                             double* records);

/// Same for the 4x4-block matrix-core E-step kernel (layout: device/layout.hpp estep_mfma4_param_stride). With a `shift`
/// (the data's d-vector) the second vector of every record, -W (mu - shift), is filled too; returns whether the kernel's FOLD
/// form may use it: every entry of every W_k (mu_k - shift) finite and at most `fold_limit` in magnitude.
bool build_estep_params_mfma4(int d, int D, int K, const double* mixing, const double* means, const double* covariances,
                              const double* shift, double fold_limit, double* records);

/// M-step closing arithmetic from the all-reduced shifted statistics (device/device.hpp stats_count):
///   mean_k = shift + S1'/S0 ; cov_k = (M2' - S1' (S1'/S0)^T) / S0 + 1e-15 I ; pi_k = S0 / N.
/// Algebraically the reference's  sum_i r_ik (x_i - mean_k)(x_i - mean_k)^T / S0  (ML/EM.cpp:245-257).
void finalize_mstep(int d, int K, const double* stats, const double* shift, double n_global, double* mixing,
                    double* means, double* covariances);

/// Diagonal-covariance extension: K_padded records [mean(D) | 1/sigma^2 (D) | log(pi) - sum log sigma | two-op flag] and the
/// dimension-major trailer [1/sigma | -(mean - shift)/sigma] behind them (layout.hpp diag_param_stride / diag_param_doubles)
/// from variances[K*d]; sigma_j = sqrt(var_j) is the diagonal Cholesky factor, 1/sigma^2 = (1/sigma)/sigma what
/// llt.solve(I) yields for it (ML/EM.cpp:279-285 restricted to a diagonal matrix).
/// `records` receives K_padded >= K records; those beyond K are neutral (coef = -inf: log-density -inf, responsibility 0).
void build_diag_params(int d, int D, int K, int K_padded, const double* mixing, const double* means, const double* variances,
                       const double* shift, double* records);

/// Diagonal closing arithmetic from the all-reduced statistics [S1'(d) | S2'(d) | S0] per component:
///   mean_k = shift + S1'/S0 ; var_kj = (S2'_j - S1'_j (S1'_j/S0)) / S0 + 1e-15 ; pi_k = S0 / N   (ML/EM.cpp:242-257, diagonal).
void finalize_mstep_diag(int d, int K, const double* stats, const double* shift, double n_global, double* mixing,
                         double* means, double* variances);

/// Tells the host-side math how many ranks share this node, so that the OpenMP teams of the per-component
/// factorizations together stay within the host's cores (MLHIP_HOST_THREADS overrides the per-rank thread count).
void set_host_ranks(int local_ranks);

}  // namespace host
}  // namespace mlhip
