// EM family of the C ABI: E-step / statistics / closing orchestration of the gfx950 kernels per step (closing on the host), full
// and diagonal covariances; the whole loop of EM::fit (mlhip_em_iterate: closing on the device) lives in em_loop.cpp.
#include "internal.hpp"

namespace mlhip_rt {


void ensure_em_workspace(mlhip_data* dt, int K)
{
    mlhip_ctx* ctx = dt->ctx;
    if (dt->em_K == K) return;
    dt->have_estep = false;
    dt->ldr = dt->n_pad;
    dt->lw.reserve(sizeof(double) * dt->ldr * K);
    dt->lse.reserve(sizeof(double) * dt->n_pad);
    dt->ll_partials.reserve(sizeof(double) * kMaxLlPartials);
    size_t ps = (size_t)estep_param_stride(dt->D) * K * sizeof(double);
#ifdef MLHIP_EXPERIMENTS
    if (estep_mfma_supported(dt->D)) ps = std::max(ps, (size_t)estep_mfma_param_stride(dt->D) * K * sizeof(double));
#endif
    if (estep_mfma4_supported(dt->D)) ps = std::max(ps, (size_t)estep_mfma4_param_stride(dt->D) * K * sizeof(double));
    dt->params_dev.reserve(ps);
    dt->params_host.reserve(ps);
    dt->partials.reserve(sizeof(double) * em_mstats_scratch_doubles(dt->d, K, ctx->num_cus));
    const size_t sb = sizeof(double) * ((size_t)K * stats_count(dt->d) + 1);
    dt->stats_dev.reserve(sb);
    dt->stats_host.reserve(sb);
    dt->em_K = K;
}


/// Builds the per-component records for the E-step kernel that fits (d, env) and uploads them to params_dev.
void prepare_estep(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs, DevBuf* target)
{
    mlhip_ctx* ctx = dt->ctx;
    ensure_em_workspace(dt, K);
    if (!target) target = &dt->params_dev;
    {   // (params_dev / params_next are swapped by mlhip_em_iterate and may have been sized for diagonal records)
        size_t ps = (size_t)estep_param_stride(dt->D) * K * sizeof(double);
        if (estep_mfma4_supported(dt->D)) ps = std::max(ps, (size_t)estep_mfma4_param_stride(dt->D) * K * sizeof(double));
#ifdef MLHIP_EXPERIMENTS
        if (estep_mfma_supported(dt->D)) ps = std::max(ps, (size_t)estep_mfma_param_stride(dt->D) * K * sizeof(double));
#endif
        target->reserve(ps);
        dt->params_host.reserve(ps);
    }
    // d in 12..128: 4x4-block triangular matrix-core kernel (mfma4). For d <= 32, MLHIP_ESTEP=valu selects the scalar-fed
    // VALU kernel (the only one below d = 12) and, in a `make EXPERIMENTS=1` build, MLHIP_ESTEP=mfma16 the 16x16x4
    // block-triangular one, for A/B runs.
    bool use_mfma = false, use_mfma4 = estep_mfma4_supported(dt->D);
    if (dt->D <= kRegDim) {
        if (const char* e = std::getenv("MLHIP_ESTEP")) {
            if (std::strcmp(e, "valu") == 0) use_mfma4 = false;
#ifdef MLHIP_EXPERIMENTS
            if (std::strcmp(e, "mfma16") == 0 && estep_mfma_supported(dt->D)) { use_mfma4 = false; use_mfma = true; }
#endif
        }
    }
    dt->estep_fold = false;
    // d > 64: the K factorizations on the device (em_close_big.hip launch_em_records_big -- the closing arithmetic's kernels, started
    // from the given covariances; the host's operations in the host's order, so the records are the host builders' except through
    // log()). On the host they were 60 ms at d = 1024, K = 4 -- once per fit, but a short fit is a few iterations. MLHIP_DEVICE_CLOSE=0
    // (or MLHIP_DEVICE_RECORDS=0 for this step alone): the host builders.
    const bool records_on_device = [] {                  // (read per call: tests switch it; MLHIP_DEVICE_RECORDS=0/1 decides for the records alone)
        const char* r = std::getenv("MLHIP_DEVICE_RECORDS");
        const char* e = r && *r ? r : std::getenv("MLHIP_DEVICE_CLOSE");
        return !(e && e[0] == '0');
    }();
    if (records_on_device && em_close_big_supported(dt->d) && !use_mfma) {
        const int d = dt->d;
        const size_t n_par = (size_t)K * ((size_t)d * d + d + 1);
        dt->close_work.reserve(sizeof(double) * em_close_work_doubles(d, K));
        double* area = em_close_big_param_area(dt->close_work.as<double>(), d, K);
        // (straight from the caller's arrays: the covariances alone are K d^2 doubles -- 0.5 GB at K = 64, d = 1024 --, no pinned copy of that)
        HIP_CHECK(hipMemcpyAsync(area, mixing, sizeof(double) * K, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(area + K, means, sizeof(double) * K * d, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(area + K + (size_t)K * d, covs, sizeof(double) * K * d * d, hipMemcpyHostToDevice, ctx->stream));
        CloseArgs ca{};
        ca.K = K; ca.d = d; ca.D = dt->D; ca.shift = dt->shift_dev.as<double>();
        ca.layout = use_mfma4 ? 2 : 0;
        ca.mixing = area; ca.means = area + K; ca.covs = area + K + (size_t)K * d;
        ca.records = target->as<double>();
        ca.info = area + n_par;
        ca.work = dt->close_work.as<double>();
        launch_em_records_big(ca, ctx->stream);
        HIP_CHECK(hipGetLastError());
        ctx->sync();                                     // (the caller's arrays may change once this returns)
    } else if (use_mfma4) {
        // FOLD form (no per-component mean subtraction in the kernel) while every |W_k (mu_k - shift)| is small enough for
        // the parity tolerances; the exact form otherwise. Every rank decides from the same parameters. MLHIP_ESTEP_FOLD=0: off.
        static const bool fold_allowed = [] { const char* e = std::getenv("MLHIP_ESTEP_FOLD"); return !(e && e[0] == '0'); }();
        const bool try_fold = fold_allowed && dt->D <= kRegDim;
        dt->estep_fold = host::build_estep_params_mfma4(dt->d, dt->D, K, mixing, means, covs, try_fold ? dt->shift.data() : nullptr,
                                                        kEstepFoldLimit, dt->params_host.as<double>());
        HIP_CHECK(hipMemcpyAsync(target->p, dt->params_host.p, sizeof(double) * estep_mfma4_param_stride(dt->D) * K,
                                 hipMemcpyHostToDevice, ctx->stream));
#ifdef MLHIP_EXPERIMENTS
    } else if (use_mfma) {
        host::build_estep_params_mfma(dt->d, dt->D, K, mixing, means, covs, dt->params_host.as<double>());
        HIP_CHECK(hipMemcpyAsync(target->p, dt->params_host.p, sizeof(double) * estep_mfma_param_stride(dt->D) * K,
                                 hipMemcpyHostToDevice, ctx->stream));
#endif
    } else {
        host::build_estep_params(dt->d, dt->D, K, mixing, means, covs, dt->params_host.as<double>());
        HIP_CHECK(hipMemcpyAsync(target->p, dt->params_host.p, sizeof(double) * estep_param_stride(dt->D) * K,
                                 hipMemcpyHostToDevice, ctx->stream));
    }
    dt->estep_variant = use_mfma4 ? 2 : (use_mfma ? 1 : 0);
}


/// E-step kernel on the records in params_dev: fills lw and -- unless the statistics kernel is going to normalise the
/// log-responsibilities itself (`with_lse` false, matrix-core kernel only) -- lse and the log-likelihood partials.
void launch_estep(mlhip_data* dt, int K, bool with_lse, const DevBuf* records, int fold)
{
    mlhip_ctx* ctx = dt->ctx;
    EstepArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.D = dt->D;
    a.params = (records ? records : &dt->params_dev)->as<double>(); a.K = K;
    a.lw = dt->lw.as<double>(); a.ldr = dt->ldr; a.lse = dt->lse.as<double>();
    a.ll_partials = dt->ll_partials.as<double>(); a.n_ll_partials = kMaxLlPartials;
    a.shift = dt->shift_dev.as<double>(); a.fold = (fold < 0 ? dt->estep_fold : fold != 0) ? 1 : 0;
    a.with_lse = (with_lse || dt->estep_variant != 2) ? 1 : 0;
    a.num_cus = ctx->num_cus;
    a.scratch = dt->partials.as<double>(); a.scratch_doubles = dt->partials.bytes / sizeof(double);   // (written by the statistics kernel AFTER the E-step, on the same stream)
    int grid = 0;
    ctx->timed("em_estep", [&] {
        if (dt->estep_variant == 2) {
            grid = -1;
#ifdef MLHIP_EXPERIMENTS
            // component-stationary form (experiments/em_estep_cs.hip: W blocks in registers, samples from LDS): measured slower
            // than the kernel below; MLHIP_ESTEP_CS=1 selects it for A/B runs
            static const bool cs = [] { const char* e = std::getenv("MLHIP_ESTEP_CS"); return e && e[0] == '1'; }();
            if (cs && a.fold && !a.with_lse && em_estep_cs_supported(a.D, K)) grid = launch_em_estep_cs(a, ctx->num_cus, ctx->stream);
#endif
            if (grid < 0) grid = launch_em_estep_mfma4(a, ctx->num_cus, ctx->stream);
        }
#ifdef MLHIP_EXPERIMENTS
        else if (dt->estep_variant == 1) grid = launch_em_estep_mfma(a, ctx->num_cus, ctx->stream);
#endif
        else grid = launch_em_estep(a, ctx->stream);
    });
    if (grid < 0) throw Unsupported("E-step kernel not instantiated for this dimension");
    HIP_CHECK(hipGetLastError());
    dt->n_ll = grid;
    dt->have_estep = true;
    dt->lw_valid = true;
}


void run_estep(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs, bool with_lse)
{
    dt->diag_step = false;
    prepare_estep(dt, K, mixing, means, covs);
    launch_estep(dt, K, with_lse);
}


/// After a fused step only lse exists on the device; whoever needs the log-responsibility block (labels,
/// responsibilities, a separate M-step, the refinement pass) gets it rebuilt from the same parameter records.
void ensure_lw(mlhip_data* dt, int K)
{
    if (!dt->have_estep || dt->lw_valid) return;
    if (dt->diag_step) {
        // params_dev holds diagonal records: expand the same parameters to full (diagonal) covariances for the E-step kernel
        const int d = dt->d;
        std::vector<double> covs((size_t)K * d * d, 0.0);
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < d; ++j) covs[(size_t)k * d * d + (size_t)j * d + j] = dt->diag_vars[(size_t)k * d + j];
        prepare_estep(dt, K, dt->diag_mixing.data(), dt->diag_means.data(), covs.data());
        dt->diag_step = false;
    }
    launch_estep(dt, K);
}


/// All-reduces the reduced statistics buffer [K*F stats, ll_sum] and leaves it in stats_host.
void collect_stats(mlhip_data* dt, int K, size_t count)
{
    mlhip_ctx* ctx = dt->ctx;
    if (!count) count = (size_t)K * stats_count(dt->d) + 1;
    if (ctx->reduce_fn && ctx->reduce_on_device) {
        ctx->reduce_device(dt->stats_dev.as<double>(), count);
    }
    HIP_CHECK(hipMemcpyAsync(dt->stats_host.p, dt->stats_dev.p, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->reduce_fn && !ctx->reduce_on_device) {
        if (ctx->reduce_fn(ctx->reduce_user, dt->stats_host.as<double>(), count, 0, ctx->stream) != 0)
            throw hook_failure();
    }
}


/// One EM iteration's device work in a single kernel where the shape allows (d <= 8, K <= 32 or d <= 4, K <= 64: em_fused_small.hip): no
/// N x K block in HBM. MLHIP_FUSED=0 keeps the two-kernel path. Returns false when the shape is not covered.
bool fused_step_applies(const mlhip_data* dt, int K)
{
    const char* env = std::getenv("MLHIP_FUSED");
    return !(env && env[0] == '0') && mstats::em_fused_supported(dt->d, K);
}


/// The fused kernel + reduction on the records already in params_dev; statistics end in stats_dev (and, with `collect`, all-
/// reduced in stats_host).
void launch_fused_step(mlhip_data* dt, int K, bool collect, const DevBuf* records)
{
    mlhip_ctx* ctx = dt->ctx;
    FusedArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.d = dt->d;
    a.shift = dt->shift_dev.as<double>(); a.params = (records ? records : &dt->params_dev)->as<double>(); a.K = K;
    a.lse = dt->lse.as<double>();
    a.partials = dt->partials.as<double>(); a.partials_capacity = dt->partials.bytes / sizeof(double);
    a.ll_partials = dt->ll_partials.as<double>(); a.n_ll_partials = kMaxLlPartials;
    int grid = 0;
    ctx->timed("em_fused", [&] { grid = mstats::launch_em_fused_small(a, ctx->num_cus, ctx->stream); });
    if (grid <= 0) throw std::runtime_error("fused EM kernel launch failed");
    launch_em_reduce_blocks(a.partials, grid, mstats::em_fused_partial_rows(K), mstats::em_fused_partial_cols(dt->d), K,
                            stats_count(dt->d), a.ll_partials, grid, dt->stats_dev.as<double>(), ctx->stream);
    HIP_CHECK(hipGetLastError());
    dt->n_ll = grid;
    dt->have_estep = true;
    dt->lw_valid = false;
    dt->stats_mode = kFromLogResp;
    dt->stats_resp = dt->lw.as<double>();
    dt->stats_ld = dt->ldr;
    if (collect) collect_stats(dt, K);
}


bool run_fused_step(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs)
{
    if (!fused_step_applies(dt, K)) return false;
    dt->diag_step = false;
    prepare_estep(dt, K, mixing, means, covs);
    if (dt->estep_variant != 0) return false;            // (cannot happen for d <= 8; the fused kernel reads VALU records)
    launch_fused_step(dt, K, true);
    return true;
}


/// Runs the statistics kernel on log-responsibilities (mode kFromLogResp: the E-step's lw/lse) or on plain
/// responsibilities `resp_dev` ([K][ld_resp], ld_resp >= n_pad), all-reduces, leaves [K*F stats, ll_sum] in stats_host.
void run_mstats(mlhip_data* dt, int K, int mode, const double* resp_dev, size_t ld_resp, bool with_ll, bool collect)
{
    mlhip_ctx* ctx = dt->ctx;
    ensure_em_workspace(dt, K);
    MstatsArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.d = dt->d;
    a.shift = dt->shift_dev.as<double>();
    a.lw = (mode == kFromResp) ? resp_dev : dt->lw.as<double>();
    a.ldr = (mode == kFromResp) ? ld_resp : dt->ldr;
    a.lse = dt->lse.as<double>();
    a.K = K; a.mode = mode;
    a.partials = dt->partials.as<double>(); a.partials_capacity = dt->partials.bytes / sizeof(double);
    a.ll_partials = with_ll ? dt->ll_partials.as<double>() : nullptr;
    a.n_ll_partials = with_ll ? dt->n_ll : 0;
    a.stats = dt->stats_dev.as<double>();
    a.lse_out = dt->lse.as<double>(); a.ll_scratch = dt->ll_partials.as<double>();
    if (mode == kFromLogRespSelfNorm) {
        dt->esum.reserve(sizeof(double) * dt->n_pad);
        a.ll_out = dt->esum.as<double>();
    }
    // after a self-normalising pass lse is in HBM like after an LSE-writing E-step: a refinement pass reads it
    dt->stats_mode = mode == kFromLogRespSelfNorm ? (int)kFromLogResp : mode;
    dt->stats_resp = a.lw;
    dt->stats_ld = a.ldr;
    int rc = 0;
    ctx->timed("em_mstats", [&] { rc = launch_em_mstats(a, ctx->num_cus, ctx->stream); });
    if (rc <= 0) throw std::runtime_error("statistics kernel launch failed (plan/scratch)");
    launch_em_reduce(a, ctx->num_cus, rc, ctx->stream);
    HIP_CHECK(hipGetLastError());
    if (collect) collect_stats(dt, K);
}


double log_two_pi()
{
    static const double v = std::log(2. * 3.14159265358979323846);   // ML/EM.cpp:197
    return v;
}


double ll_from_stats(const mlhip_data* dt, int K)
{
    const double log_2_pi = log_two_pi();
    const double sum = dt->stats_host.as<double>()[(size_t)K * stats_count(dt->d)];
    return sum / (double)dt->n_global - (double)dt->d * log_2_pi / 2;
}


void check_em_args(mlhip_ctx* ctx, mlhip_data* dt, uint32_t K)
{
    require(ctx && dt, "null context or data");
    require(dt->ctx == ctx, "data belongs to another context");
    require(K >= 1, "At least one component required");
    ctx->use();
}


/// Ratio (mean offset from the shared shift)^2 / variance above which a component's covariance is recomputed about its
/// own mean. The one-GEMM statistics share one shift (the global mean), so Sigma_k = M2'/S0 - m m^T cancels
/// ~log10(ratio) digits: measured relative error ~3e-15 * ratio. 1e4 keeps every covariance within ~3e-11 of the
/// two-pass form the reference uses (ML/EM.cpp:245-250). MLHIP_REFINE_RATIO overrides; <= 0 disables the refinement.
double refine_ratio()
{
    static const double r = [] {
        const char* e = std::getenv("MLHIP_REFINE_RATIO");
        return (e && *e) ? std::atof(e) : 1e4;
    }();
    return r;
}


/// Second statistics pass for ONE component with the shift at that component's new mean (K = 1 launch of the same
/// kernels on column k of the responsibilities of the last pass), all-reduced like the first; replaces covariance k
/// (and adds the tiny mean correction). Tight clusters far from the global mean need it; the headline shapes never do.
void refine_component(mlhip_data* dt, int k, double* mean_k, double* cov_k)
{
    mlhip_ctx* ctx = dt->ctx;
    const int d = dt->d, F = stats_count(d);
    if (dt->stats_mode == kFromLogResp) ensure_lw(dt, dt->em_K);   // after a fused step the block is not in HBM yet
    dt->refine_shift.reserve(sizeof(double) * d);
    dt->refine_stats.reserve(sizeof(double) * (F + 1));
    HIP_CHECK(hipMemcpyAsync(dt->refine_shift.p, mean_k, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
    MstatsArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.d = d;
    a.shift = dt->refine_shift.as<double>();
    a.lw = dt->stats_resp + (size_t)k * dt->stats_ld; a.ldr = dt->stats_ld; a.lse = dt->lse.as<double>();
    a.K = 1; a.mode = dt->stats_mode;
    a.partials = dt->partials.as<double>(); a.partials_capacity = dt->partials.bytes / sizeof(double);
    a.ll_partials = nullptr; a.n_ll_partials = 0;
    a.stats = dt->refine_stats.as<double>();
    int rc = 0;
    ctx->timed("em_refine", [&] { rc = launch_em_mstats(a, ctx->num_cus, ctx->stream); });
    if (rc <= 0) throw std::runtime_error("statistics kernel launch failed (refinement pass)");
    launch_em_reduce(a, ctx->num_cus, rc, ctx->stream);
    HIP_CHECK(hipGetLastError());
    std::vector<double> s((size_t)F);
    if (ctx->reduce_fn && ctx->reduce_on_device) {
        if (ctx->reduce_fn(ctx->reduce_user, dt->refine_stats.as<double>(), (size_t)F, 1, ctx->stream) != 0)
            throw hook_failure();
    }
    HIP_CHECK(hipMemcpyAsync(s.data(), dt->refine_stats.p, sizeof(double) * F, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->reduce_fn && !ctx->reduce_on_device) {
        if (ctx->reduce_fn(ctx->reduce_user, s.data(), (size_t)F, 0, ctx->stream) != 0) throw hook_failure();
    }
    const double s0 = s[stats_index(d, d)];
    std::vector<double> m(d);
    for (int a2 = 0; a2 < d; ++a2) m[a2] = s[stats_index(d, a2)] / s0;          // ~0: the shift is the mean already
    for (int a2 = 0; a2 < d; ++a2)
        for (int b = 0; b <= a2; ++b) {
            const double v = (s[stats_index(a2, b)] - s[stats_index(d, a2)] * m[b]) / s0;
            cov_k[b * d + a2] = v;
            cov_k[a2 * d + b] = v;
        }
    for (int a2 = 0; a2 < d; ++a2) {
        cov_k[a2 * d + a2] += 1e-15;                                            // ML/EM.cpp:252
        mean_k[a2] += m[a2];
    }
    dt->refined_components += 1;
}


void finalize_out(mlhip_data* dt, int K, double* mixing_out, double* means_out, double* cov_out)
{
    const int d = dt->d;
    host::finalize_mstep(d, K, dt->stats_host.as<double>(), dt->shift.data(), (double)dt->n_global, mixing_out,
                         means_out, cov_out);
    const double limit = refine_ratio();
    if (!(limit > 0)) return;
    // Every rank sees the same all-reduced statistics, hence flags the same components in the same order.
    for (int k = 0; k < K; ++k) {
        const double* mu = means_out + (size_t)k * d;
        const double* cov = cov_out + (size_t)k * d * d;
        if (!(mixing_out[k] > 0) || !std::isfinite(mixing_out[k])) continue;    // empty / broken component: as the reference
        bool flag = false;
        for (int a = 0; a < d && !flag; ++a) {
            const double off = mu[a] - dt->shift[a], var = cov[a * d + a];
            if (!std::isfinite(off) || !std::isfinite(var)) { flag = false; break; }   // NaN stays NaN (ML/EM.cpp:236)
            flag = off * off > limit * var;                                      // also catches var <= 0 from cancellation
        }
        if (flag) refine_component(dt, k, means_out + (size_t)k * d, cov_out + (size_t)k * d * d);
    }
}


/// One diagonal-covariance EM iteration's device work (em_diag.hip) with the statistics shift at `shift_dev`; leaves the
/// all-reduced [K * (2d+1) statistics, ll_sum] in stats_host. The records must already be in params_dev.
void run_diag_kernel(mlhip_data* dt, int K, const double* shift_dev, bool collect, const DevBuf* records)
{
    mlhip_ctx* ctx = dt->ctx;
    DiagArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.d = dt->d;
    a.shift = shift_dev; a.params = (records ? records : &dt->params_dev)->as<double>(); a.K = K;
    a.lse = dt->lse.as<double>();
    a.partials = dt->partials.as<double>(); a.partials_capacity = dt->partials.bytes / sizeof(double);
    a.ll_partials = dt->ll_partials.as<double>(); a.n_ll_partials = kMaxLlPartials;
    a.two_op = shift_dev == dt->shift_dev.as<double>() ? 1 : 0;     // (the records' a, b are relative to the data's shift)
    int grid = 0;
    ctx->timed("em_diag", [&] { grid = mstats::launch_em_diag(a, ctx->num_cus, ctx->stream); });
    if (grid <= 0) throw std::runtime_error("diagonal EM kernel launch failed");
    launch_em_reduce_blocks(a.partials, grid, mstats::em_diag_partial_rows(K), mstats::em_diag_partial_cols(dt->d), K,
                            diag_stats_count(dt->d), a.ll_partials, grid, dt->stats_dev.as<double>(), ctx->stream);
    HIP_CHECK(hipGetLastError());
    dt->n_ll = grid;
    if (collect) collect_stats(dt, K, (size_t)K * diag_stats_count(dt->d) + 1);
}


/// K within one row-block group of the wide statistics kernel: the matrix-core E-step writes the log-responsibilities only and
/// the statistics kernel normalises them (one exp per pair in the iteration); otherwise the E-step keeps its online
/// log-sum-exp. MLHIP_SELF_NORM=0 forces the latter (A/B runs).
bool self_norm_applies(const mlhip_data* dt, int K)
{
    static const bool allowed = [] { const char* e = std::getenv("MLHIP_SELF_NORM"); return !(e && e[0] == '0'); }();
    return allowed && estep_mfma4_supported(dt->D) && !std::getenv("MLHIP_ESTEP") &&
           em_mstats_self_norm_supported(dt->d, K, dt->ctx->num_cus);
}


/// One full-covariance EM iteration with the closing arithmetic on the HOST (the body of mlhip_em_step).
void em_step_full(mlhip_data* data, int K, const double* mixing, const double* means, const double* covariances,
                  double* log_likelihood, double* mixing_out, double* means_out, double* covariances_out)
{
    PhaseTrace tr;
    if (run_fused_step(data, K, mixing, means, covariances)) {
        tr.mark("fused E+M launch+sync+D2H");
    } else {
        const bool self_norm = self_norm_applies(data, K);
        run_estep(data, K, mixing, means, covariances, !self_norm);
        tr.mark("params+launch E");
        run_mstats(data, K, self_norm && data->estep_variant == 2 ? kFromLogRespSelfNorm : kFromLogResp, nullptr, 0, true);
        tr.mark("M launch+sync+D2H");
    }
    *log_likelihood = ll_from_stats(data, K);
    finalize_out(data, K, mixing_out, means_out, covariances_out);
    tr.mark("closing arithmetic");
}


/// Sums `count` doubles at the head of stats_dev across ranks, whatever kind of hook is installed (device buffer on the
/// stream, or a host buffer: down, hook, up). No-op on a single rank.
void allreduce_stats_dev(mlhip_data* dt, size_t count)
{
    mlhip_ctx* ctx = dt->ctx;
    if (!ctx->reduce_fn) return;
    if (ctx->reduce_on_device) {
        ctx->reduce_device(dt->stats_dev.as<double>(), count);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(dt->stats_host.p, dt->stats_dev.p, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->reduce_fn(ctx->reduce_user, dt->stats_host.as<double>(), count, 0, ctx->stream) != 0)
        throw hook_failure();
    HIP_CHECK(hipMemcpyAsync(dt->stats_dev.p, dt->stats_host.p, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
}


/// Records of a diagonal-covariance parameter set -> `target` (padded to whole 16-component row blocks with neutral records).
void upload_diag_records(mlhip_data* data, int K, const double* mixing, const double* means, const double* variances, DevBuf& target)
{
    mlhip_ctx* ctx = data->ctx;
    const int KP = mstats::em_diag_partial_rows(K);
    const size_t rec_bytes = sizeof(double) * diag_param_doubles(data->D, KP);
    target.reserve(rec_bytes);
    data->params_host.reserve(rec_bytes);
    host::build_diag_params(data->d, data->D, K, KP, mixing, means, variances, data->shift.data(), data->params_host.as<double>());
    HIP_CHECK(hipMemcpyAsync(target.p, data->params_host.p, rec_bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();                                     // params_host may be rewritten right away by the caller's next upload
}


/// Same cancellation guard as the full-covariance path (refine_ratio): a component whose mean sits far from the shared shift,
/// measured in its own standard deviations, gets its variances from a second pass with the shift at its new mean (the E part of
/// that pass re-evaluates the SAME input parameters, still in params_dev).
void refine_diag(mlhip_data* data, int K, const double* mixing_out, double* means_out, double* variances_out)
{
    mlhip_ctx* ctx = data->ctx;
    const int d = data->d, F = diag_stats_count(d);
    const double limit = refine_ratio();
    if (!(limit > 0)) return;
    for (int k = 0; k < K; ++k) {
        if (!(mixing_out[k] > 0) || !std::isfinite(mixing_out[k])) continue;
        bool flag = false;
        for (int a = 0; a < d && !flag; ++a) {
            const double off = means_out[(size_t)k * d + a] - data->shift[a], var = variances_out[(size_t)k * d + a];
            if (!std::isfinite(off) || !std::isfinite(var)) { flag = false; break; }
            flag = off * off > limit * var;
        }
        if (!flag) continue;
        data->refine_shift.reserve(sizeof(double) * data->D);
        HIP_CHECK(hipMemsetAsync(data->refine_shift.p, 0, sizeof(double) * data->D, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(data->refine_shift.p, means_out + (size_t)k * d, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
        run_diag_kernel(data, K, data->refine_shift.as<double>());
        const double* s = data->stats_host.as<double>() + (size_t)k * F;
        const double s0 = s[2 * d];
        for (int a = 0; a < d; ++a) {
            const double m = s[a] / s0;                                      // ~0: the shift is the mean already
            variances_out[(size_t)k * d + a] = (s[d + a] - s[a] * m) / s0 + 1e-15;
            means_out[(size_t)k * d + a] += m;
        }
        data->refined_components += 1;
    }
}


void em_step_diag(mlhip_data* data, int K, const double* mixing, const double* means, const double* variances,
                  double* log_likelihood, double* mixing_out, double* means_out, double* variances_out)
{
    const int d = data->d;
    if (!mstats::em_diag_supported(d, K)) {
        // Shapes the one-kernel diagonal iteration is not built for (d > 32 or K > 64): the same iteration through the
        // full-covariance kernels on diagonal matrices -- the E-step's Cholesky of a diagonal matrix is its square root, and
        // the diagonal of the M-step's full covariance IS the diagonal-mode variance (ML/EM.cpp:245-257 entry by entry); the
        // off-diagonal sums are computed and dropped. Slower than it could be, never refused.
        std::vector<double> cov((size_t)K * d * d, 0.0), cov_out((size_t)K * d * d);
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < d; ++j) cov[((size_t)k * d + j) * d + j] = variances[(size_t)k * d + j];
        em_step_full(data, K, mixing, means, cov.data(), log_likelihood, mixing_out, means_out, cov_out.data());
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < d; ++j) variances_out[(size_t)k * d + j] = cov_out[((size_t)k * d + j) * d + j];
        return;
    }
    ensure_em_workspace(data, K);
    // keep the input parameters: labels / responsibilities are produced from them on demand (ensure_lw)
    data->diag_mixing.assign(mixing, mixing + K);
    data->diag_means.assign(means, means + (size_t)K * d);
    data->diag_vars.assign(variances, variances + (size_t)K * d);
    upload_diag_records(data, K, mixing, means, variances, data->params_dev);
    run_diag_kernel(data, K, data->shift_dev.as<double>());
    data->have_estep = true;
    data->lw_valid = false;
    data->diag_step = true;
    const int F = diag_stats_count(d);
    const double* st = data->stats_host.as<double>();
    *log_likelihood = st[(size_t)K * F] / (double)data->n_global - (double)d * log_two_pi() / 2;   // ML/EM.cpp:197-198, 211
    host::finalize_mstep_diag(d, K, st, data->shift.data(), (double)data->n_global, mixing_out, means_out, variances_out);
    refine_diag(data, K, mixing_out, means_out, variances_out);
}


}  // namespace mlhip_rt

extern "C" {


int mlhip_em_expectation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* mixing, const double* means,
                         const double* covariances, double* log_likelihood)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::em_expectation(ctx, data, K, mixing, means, covariances, log_likelihood); return; }
        check_em_args(ctx, data, K);
        require(mixing && means && covariances && log_likelihood, "null argument");
        run_estep(data, (int)K, mixing, means, covariances);
        double* slot = data->stats_dev.as<double>() + (size_t)K * stats_count(data->d);
        launch_ll_reduce(data->ll_partials.as<double>(), data->n_ll, slot, ctx->stream);
        HIP_CHECK(hipGetLastError());
        double* host_slot = data->stats_host.as<double>() + (size_t)K * stats_count(data->d);
        HIP_CHECK(hipMemcpyAsync(host_slot, slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        ctx->allreduce_host(host_slot, 1);
        *log_likelihood = ll_from_stats(data, (int)K);
    });
}

int mlhip_em_maximisation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* mixing_out, double* means_out,
                          double* covariances_out)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::em_maximisation(ctx, data, K, 0, nullptr, 0, nullptr, mixing_out, means_out, covariances_out); return; }
        check_em_args(ctx, data, K);
        require(mixing_out && means_out && covariances_out, "null argument");
        require(data->have_estep && data->em_K == (int)K, "no E-step results on the device for this K");
        ensure_lw(data, (int)K);
        run_mstats(data, (int)K, kFromLogResp, nullptr, 0, true);
        finalize_out(data, (int)K, mixing_out, means_out, covariances_out);
    });
}

int mlhip_em_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* mixing, const double* means,
                  const double* covariances, double* log_likelihood, double* mixing_out, double* means_out,
                  double* covariances_out)
{
    return guarded([&] {
        if (ctx && ctx->group) {
            grp::em_step(ctx, data, K, false, mixing, means, covariances, log_likelihood, mixing_out, means_out, covariances_out);
            return;
        }
        check_em_args(ctx, data, K);
        require(mixing && means && covariances && log_likelihood && mixing_out && means_out && covariances_out, "null argument");
        em_step_full(data, (int)K, mixing, means, covariances, log_likelihood, mixing_out, means_out, covariances_out);
    });
}

int mlhip_em_step_diag(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* mixing, const double* means,
                       const double* variances, double* log_likelihood, double* mixing_out, double* means_out,
                       double* variances_out)
{
    return guarded([&] {
        if (ctx && ctx->group) {
            grp::em_step(ctx, data, K, true, mixing, means, variances, log_likelihood, mixing_out, means_out, variances_out);
            return;
        }
        check_em_args(ctx, data, K);
        require(mixing && means && variances && log_likelihood && mixing_out && means_out && variances_out, "null argument");
        em_step_diag(data, (int)K, mixing, means, variances, log_likelihood, mixing_out, means_out, variances_out);
    });
}

int mlhip_em_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, int covariance_type, double* mixing, double* means,
                     double* covariances, uint32_t max_steps, double absolute_tolerance, double relative_tolerance,
                     uint32_t* steps_done, int* converged, double* log_likelihood, double* log_likelihood_history)
{
    return guarded([&] {
        if (ctx && ctx->group) {
            require(max_steps >= 1, "at least one step required");
            if (absolute_tolerance < 0 || relative_tolerance < 0) throw DomainError("negative tolerance");
            grp::em_iterate(ctx, data, K, covariance_type, mixing, means, covariances, max_steps, absolute_tolerance, relative_tolerance,
                            steps_done, converged, log_likelihood, log_likelihood_history);
            return;
        }
        check_em_args(ctx, data, K);
        require(mixing && means && covariances && steps_done && converged && log_likelihood, "null argument");
        require(covariance_type == MLHIP_COVARIANCE_FULL || covariance_type == MLHIP_COVARIANCE_DIAGONAL, "bad covariance_type");
        require(max_steps >= 1, "at least one step required");
        if (absolute_tolerance < 0 || relative_tolerance < 0) throw DomainError("negative tolerance");
        const bool diag = covariance_type == MLHIP_COVARIANCE_DIAGONAL;
        em_iterate(data, (int)K, diag, mixing, means, covariances, max_steps, absolute_tolerance, relative_tolerance, steps_done,
                   converged, log_likelihood, log_likelihood_history);
        const size_t cov_doubles = (size_t)K * data->d * (diag ? 1 : data->d);
        ctx->check_ranks_agree("the EM parameters", {{mixing, K}, {means, (size_t)K * data->d}, {covariances, cov_doubles}, {log_likelihood, 1}});
    });
}

int mlhip_em_maximisation_from(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* resp, int64_t ldr,
                               double* mixing_out, double* means_out, double* covariances_out)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::em_maximisation(ctx, data, K, 1, resp, ldr, nullptr, mixing_out, means_out, covariances_out); return; }
        check_em_args(ctx, data, K);
        require((resp || data->n == 0) && mixing_out && means_out && covariances_out, "null argument");   // (an empty shard has no rows)
        require(ldr >= (int64_t)data->n, "ldr must be >= n_local");
        ensure_em_workspace(data, (int)K);
        data->resp_dev.reserve(sizeof(double) * data->ldr * K);
        HIP_CHECK(hipMemsetAsync(data->resp_dev.p, 0, sizeof(double) * data->ldr * K, ctx->stream));
        if (data->n)
            HIP_CHECK(hipMemcpy2DAsync(data->resp_dev.p, sizeof(double) * data->ldr, resp, sizeof(double) * ldr,
                                       sizeof(double) * data->n, K, hipMemcpyHostToDevice, ctx->stream));
        run_mstats(data, (int)K, kFromResp, data->resp_dev.as<double>(), data->ldr, false);
        finalize_out(data, (int)K, mixing_out, means_out, covariances_out);
    });
}

int mlhip_em_maximisation_from_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* labels,
                                      double* mixing_out, double* means_out, double* covariances_out)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::em_maximisation(ctx, data, K, 2, nullptr, 0, labels, mixing_out, means_out, covariances_out); return; }
        check_em_args(ctx, data, K);
        require((labels || data->n == 0) && mixing_out && means_out && covariances_out, "null argument");
        ensure_em_workspace(data, (int)K);
        data->labels_dev.reserve(sizeof(uint32_t) * data->n_pad);
        if (data->n)
            HIP_CHECK(hipMemcpyAsync(data->labels_dev.p, labels, sizeof(uint32_t) * data->n, hipMemcpyHostToDevice, ctx->stream));
        // One-hot responsibilities are materialised in the (still unused) log-responsibility buffer of the workspace.
        data->have_estep = false;
        launch_fill_responsibilities(data->labels_dev.as<uint32_t>(), data->n, (int)K, data->lw.as<double>(), data->ldr, ctx->stream);
        run_mstats(data, (int)K, kFromResp, data->lw.as<double>(), data->ldr, false);
        finalize_out(data, (int)K, mixing_out, means_out, covariances_out);
    });
}

int mlhip_em_responsibilities_rows(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint64_t first_row, uint64_t n_rows, double* resp,
                                   int64_t ldr)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::em_responsibilities(ctx, data, K, resp, ldr, first_row, n_rows); return; }
        check_em_args(ctx, data, K);
        require(first_row <= data->n && n_rows <= data->n - first_row, "row range beyond this block");
        require(resp || n_rows == 0, "null argument");
        require(ldr >= (int64_t)n_rows, "ldr must be >= the number of rows");
        require(data->have_estep && data->em_K == (int)K, "no E-step results on the device for this K");
        if (!n_rows) return;
        ensure_lw(data, (int)K);
        const size_t ldo = (size_t)padded_samples(n_rows);
        data->resp_dev.reserve(sizeof(double) * ldo * K);
        RespArgs a{data->lw.as<double>() + first_row, data->ldr, data->lse.as<double>() + first_row, (uint32_t)n_rows, (int)K,
                   data->resp_dev.as<double>(), ldo, nullptr};
        launch_em_responsibilities(a, ctx->stream);
        HIP_CHECK(hipGetLastError());
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(resp), sizeof(double) * ldr, data->resp_dev.as<char>(), sizeof(double) * ldo,
                         sizeof(double) * n_rows, K);
    });
}

int mlhip_em_responsibilities(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* resp, int64_t ldr)
{
    if (!data) return mlhip_em_responsibilities_rows(ctx, data, K, 0, 0, resp, ldr);      // (reports the null argument)
    const uint64_t n = data->parts.empty() ? (uint64_t)data->n : data->n_global;
    return mlhip_em_responsibilities_rows(ctx, data, K, 0, n, resp, ldr);
}

int mlhip_em_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint32_t* labels)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::em_labels(ctx, data, K, labels); return; }
        check_em_args(ctx, data, K);
        require(labels || data->n == 0, "null argument");
        require(data->have_estep && data->em_K == (int)K, "no E-step results on the device for this K");
        ensure_lw(data, (int)K);
        data->labels_dev.reserve(sizeof(uint32_t) * data->n_pad);
        RespArgs a{data->lw.as<double>(), data->ldr, data->lse.as<double>(), data->n, (int)K, nullptr, 0,
                   data->labels_dev.as<uint32_t>()};
        launch_em_responsibilities(a, ctx->stream);
        HIP_CHECK(hipGetLastError());
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(labels), 0, data->labels_dev.as<char>(), 0, sizeof(uint32_t) * data->n, 1);
    });
}

int mlhip_em_statistics_count(uint32_t d, uint32_t* count_per_component)
{
    return guarded([&] {
        require(d >= 1 && count_per_component, "bad argument");
        *count_per_component = (uint32_t)stats_count((int)d);
    });
}

int mlhip_em_finalize_statistics(uint32_t d, uint32_t K, const double* statistics, const double* shift, double n_global,
                                 double* mixing_out, double* means_out, double* covariances_out)
{
    return guarded([&] {
        require(d >= 1 && K >= 1 && statistics && shift && mixing_out && means_out && covariances_out, "bad argument");
        host::finalize_mstep((int)d, (int)K, statistics, shift, n_global, mixing_out, means_out, covariances_out);
    });
}

int mlhip_process_covariance(uint32_t d, const double* covariance, double* inverse, double* sqrt_det)
{
    return guarded([&] {
        require(d >= 1 && covariance && inverse && sqrt_det, "bad argument");
        host::process_covariance((int)d, covariance, inverse, sqrt_det);
    });
}

int mlhip_em_plan(const mlhip_data* data, uint32_t K, uint32_t* flags)
{
    return guarded([&] {
        require(data && flags && K >= 1, "null argument");
        if (!data->parts.empty()) data = data->parts[0];        // (a group's block: every shard takes the same plan)
        uint32_t f = 0;
        const bool matrix = estep_mfma4_supported(data->D) && !(data->D <= kRegDim && std::getenv("MLHIP_ESTEP"));
        if (fused_step_applies(data, (int)K)) f |= MLHIP_PLAN_FUSED;
        else {
            if (matrix) f |= MLHIP_PLAN_MATRIX_ESTEP;
            if (matrix && self_norm_applies(data, (int)K)) f |= MLHIP_PLAN_SELF_NORM;
        }
        *flags = f;
    });
}

}  // extern "C"
