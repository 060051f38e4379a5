// mlhip_em_iterate: the loop of EM::fit (ML/EM.cpp:143-170) with everything between two convergence tests on the device -- E-step,
// statistics, all-reduce, closing arithmetic + the next E-step's records (em_close.hip); per iteration the host reads back 1 + 2K
// doubles (log-likelihood sum, refinement flags, FOLD criterion) and decides.
//
// Three pieces, kept apart:
//   * EmLoop::launch(i)        -- ONE iteration's device work, nothing waits. Iteration i reads the records in ring slot i % 3 and
//                                 writes the new parameters into pack (i + 1) % 3 and the next records into ring slot (i + 1) % 3:
//                                 the same indexing for every loop policy, no buffer swapping while the loop runs;
//   * the loop policies        -- run_synchronous (launch, wait, test: every shape), run_lagged (iteration i + 1 is launched before
//                                 the host looks at iteration i: short iterations), run_resident (the WHOLE loop in one launch of
//                                 resident workgroups, em_resident.hip: fits whose iteration is a few microseconds),
//                                 host_closing_loop (d > 1024, MLHIP_DEVICE_CLOSE=0);
//   * close_on_host(i)         -- the one iteration a refinement flag (far, tight component) sends through the per-step arithmetic.
// finish() leaves the device state as the per-step entry points expect it (records of the LAST E-step in params_dev).
#include "internal.hpp"

namespace mlhip_rt {
namespace {

/// ML/EM.cpp:161-168: from the second trip on, |ll - ll_old| < atol + rtol max(|ll_old|, |ll|) stops the loop.
struct ConvergenceTest {
    double atol, rtol;
    uint32_t* steps_done;
    int* converged;
    double* log_likelihood;
    double* history;
    double old_ll = -HUGE_VAL;
    bool operator()(uint32_t step, double ll)
    {
        if (history) history[step] = ll;
        *log_likelihood = ll;
        *steps_done = step + 1;
        if (step > 0) {
            const double change = std::fabs(ll - old_ll);
            if (change < atol + rtol * std::max(std::fabs(old_ll), std::fabs(ll))) { *converged = 1; return true; }
        }
        old_ll = ll;
        return false;
    }
};

bool env_allows(const char* name)
{
    const char* e = std::getenv(name);
    return !(e && e[0] == '0');
}

struct EmLoop {
    mlhip_data* data;
    mlhip_ctx* ctx;
    const int K, d;
    const bool diag;
    double *mixing, *means, *covs;              // the caller's arrays (start -> result)
    // ---- plan of an iteration
    bool fused = false, self_norm = false;
    int resident_grid = 0;                      // > 0: the whole loop can run in one launch of this many resident workgroups
    bool resident_gave_up = false;
    bool info_pinned = false;                   // full covariances: the closing kernel writes its info block into pinned host memory
    bool pack_pinned = false;                   // diagonal mode: its whole (small) pack lives there
    size_t n_cov = 0, F = 0, n_info = 0, n_pack = 0;
    double refine_limit = 0;
    bool fold_allowed = true;
    // ---- ring: slot s holds the records R_i of every iteration i with i % 3 == s, pack s the parameters P_i
    DevBuf* rec[3] = {nullptr, nullptr, nullptr};
    bool fold[3] = {false, false, false};       // matrix-core E-step: whether the records of a slot are evaluated in FOLD form
    std::vector<double> shadow[3];              // diagonal mode: host copy of P_i (ensure_lw rebuilds the block from an E-step's inputs)
    uint32_t launched = 0;                      // iterations launched so far

    EmLoop(mlhip_data* dt, int K_, bool diag_, double* mixing_, double* means_, double* covs_)
        : data(dt), ctx(dt->ctx), K(K_), d(dt->d), diag(diag_), mixing(mixing_), means(means_), covs(covs_) {}

    double* pack_base(int slot) const { return pack_pinned ? data->it_info_slot[slot].as<double>() : data->it_pack[slot].as<double>(); }
    double* pack_mixing(int slot) const { return pack_base(slot) + n_info; }
    double* pack_means(int slot) const { return pack_mixing(slot) + K; }
    double* pack_covs(int slot) const { return pack_means(slot) + (size_t)K * d; }
    const double* host_info(int slot) const { return data->it_info_slot[slot].as<double>(); }

    /// Records R_0 of the caller's parameters into ring slot 0 (= params_dev), buffers of the packs and the ring.
    /// Returns false when the shape / build has no device closing (the caller runs host_closing_loop).
    bool prepare()
    {
        ensure_em_workspace(data, K);
        if (!env_allows("MLHIP_DEVICE_CLOSE") || !em_close_supported(d) || (diag && !mstats::em_diag_supported(d, K))) return false;
        if (!diag) {
            prepare_estep(data, K, mixing, means, covs);           // -> params_dev, estep_variant, estep_fold
            if (data->estep_variant == 1) return false;            // (experimental record layout: host closing only)
        }
        n_cov = diag ? (size_t)K * d : (size_t)K * d * d;
        F = diag ? diag_stats_count(d) : stats_count(d);
        n_info = em_close_info_doubles(K);
        n_pack = n_info + K + (size_t)K * d + n_cov;
        const bool pinned = !(ab_env("MLHIP_INFO_PINNED") && ab_env("MLHIP_INFO_PINNED")[0] == '0');
        info_pinned = pinned && !diag;
        pack_pinned = pinned && diag;
        for (int s = 0; s < 3; ++s) {
            data->it_pack[s].reserve(sizeof(double) * n_pack);
            data->it_info_slot[s].reserve(sizeof(double) * n_pack);
            if (!data->it_event[s]) HIP_CHECK(hipEventCreateWithFlags(&data->it_event[s], hipEventDisableTiming));
        }
        if (const size_t w = em_close_work_doubles(d, K); w > 0 && !diag) data->close_work.reserve(sizeof(double) * w);   // (before prepare_estep: it uses the same block)
        rec[0] = &data->params_dev; rec[1] = &data->params_next; rec[2] = &data->params_prev;
        if (diag) {
            for (DevBuf* r : rec) upload_diag_records(data, K, mixing, means, covs, *r);   // (the neutral padding records live in all)
            shadow[0].assign(mixing, mixing + K);
            shadow[0].insert(shadow[0].end(), means, means + (size_t)K * d);
            shadow[0].insert(shadow[0].end(), covs, covs + n_cov);
        } else {
            rec[1]->reserve(rec[0]->bytes);
            rec[2]->reserve(rec[0]->bytes);
        }
        data->diag_step = diag;
        fused = !diag && data->estep_variant == 0 && fused_step_applies(data, K);
        self_norm = !diag && !fused && data->estep_variant == 2 && self_norm_applies(data, K);
        // Single rank, a shape of the vector-unit E+M form with at most one workgroup per CU: the whole loop in one launch.
        // MLHIP_RESIDENT=0: off.
        resident_grid = 0;
        if (fused && !ctx->reduce_fn && ctx->world_size <= 1 && env_allows("MLHIP_RESIDENT")) {
            const FusedArgs fa = fused_args(rec[0]);
            const int g = mstats::em_fused_valu_small_grid(fa, ctx->num_cus);
            if (g > 0 && mstats::em_resident_supported(d, K, g, ctx->num_cus)) resident_grid = g;
        }
        fold_allowed = env_allows("MLHIP_ESTEP_FOLD");
        refine_limit = refine_ratio();
        fold[0] = data->estep_fold;
        return true;
    }

    /// Iteration i's device work: E-step + statistics from ring slot i % 3, all-reduce, closing arithmetic into pack (i + 1) % 3 and the
    /// next records into ring slot (i + 1) % 3, the info block on its way to the host, an event behind it. Nothing here waits.
    void launch(uint32_t i)
    {
        const int in = (int)(i % 3), out = (int)((i + 1) % 3);
        if (diag) {
            run_diag_kernel(data, K, data->shift_dev.as<double>(), false, rec[in]);
        } else if (fused) {
            launch_fused_step(data, K, false, rec[in]);
        } else {
            launch_estep(data, K, !self_norm, rec[in], fold[in] ? 1 : 0);
            run_mstats(data, K, self_norm ? kFromLogRespSelfNorm : kFromLogResp, nullptr, 0, true, false);
        }
        data->have_estep = true;
        data->lw_valid = !(diag || fused);
        allreduce_stats_dev(data, (size_t)K * F + 1);
        CloseArgs ca{};
        ca.stats = data->stats_dev.as<double>(); ca.K = K; ca.d = d; ca.D = data->D;
        ca.shift = data->shift_dev.as<double>(); ca.n_global = (double)data->n_global;
        ca.layout = data->estep_variant; ca.refine_limit = refine_limit;
        ca.mixing = pack_mixing(out); ca.means = pack_means(out); ca.covs = pack_covs(out);
        ca.records = rec[out]->as<double>();
        ca.info = info_pinned ? data->it_info_slot[out].as<double>() : pack_base(out);
        ca.work = data->close_work.as<double>();
        ctx->timed("em_close", [&] { if (diag) launch_em_close_diag(ca, ctx->stream); else launch_em_close(ca, ctx->stream); });
        HIP_CHECK(hipGetLastError());
        if (!info_pinned && !pack_pinned)
            HIP_CHECK(hipMemcpyAsync(data->it_info_slot[out].p, data->it_pack[out].p, sizeof(double) * (diag ? n_pack : n_info),
                                     hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipEventRecord(data->it_event[out], ctx->stream));
        launched = i + 1;
    }

    /// The fused kernel's arguments on the records of `records` (launch_fused_step builds the same).
    FusedArgs fused_args(const DevBuf* records) const
    {
        FusedArgs a{};
        a.xt = data->xt.as<double>(); a.ldx = data->ldx; a.n = data->n; a.d = d;
        a.shift = data->shift_dev.as<double>(); a.params = records->as<double>(); a.K = K;
        a.lse = data->lse.as<double>();
        a.partials = data->partials.as<double>(); a.partials_capacity = data->partials.bytes / sizeof(double);
        a.ll_partials = data->ll_partials.as<double>(); a.n_ll_partials = kMaxLlPartials;
        return a;
    }

    /// Resident policy (em_resident.hip): ONE launch runs the iterations 0 .. until the convergence test of ML/EM.cpp:161-168 fires
    /// on the device, max_steps is reached or a refinement flag comes up; the host waits once and reads [status, iterations, converged]
    /// and the log-likelihood history from pinned memory. The kernel evaluates the same pass, sums and closing arithmetic as the
    /// three launches of launch(), so everything it leaves in the ring is bit for bit what they leave. Returns true when the loop is
    /// over; false when the synchronous policy has to take over at iteration *resume_at (a refinement flag there -- or, with
    /// *resume_at = 0 and nothing consumed, a wait inside the kernel that gave up: the GPU was not ours alone).
    bool run_resident(ConvergenceTest& test, uint32_t max_steps, uint32_t* resume_at)
    {
        const size_t head = 4;                                           // result words (as doubles' worth of space), then the history
        data->it_history.reserve(sizeof(double) * (head + max_steps));
        data->it_sync.reserve(256);
        data->it_xch.reserve(sizeof(double) * mstats::em_resident_exchange_doubles(d, K, resident_grid));
        uint32_t* result = data->it_history.as<uint32_t>();
        double* history = data->it_history.as<double>() + head;
        result[0] = result[1] = result[2] = 0;
        HIP_CHECK(hipMemsetAsync(data->it_sync.p, 0, 256, ctx->stream));
        // (the exchanged values carry their iteration's number as a tag, counted from 1 in every launch: no left-over of an earlier fit may look valid)
        HIP_CHECK(hipMemsetAsync(data->it_xch.p, 0, sizeof(double) * mstats::em_resident_exchange_doubles(d, K, resident_grid), ctx->stream));
        ResidentArgs a{};
        a.xt = data->xt.as<double>(); a.ldx = data->ldx; a.n = data->n; a.d = d; a.K = K;
        a.shift = data->shift_dev.as<double>();
        for (int s = 0; s < 3; ++s) {
            a.records[s] = rec[s]->as<double>();
            a.info[s] = info_pinned ? data->it_info_slot[s].as<double>() : pack_base(s);
            a.mixing[s] = pack_mixing(s); a.means[s] = pack_means(s); a.covs[s] = pack_covs(s);
        }
        a.xch = data->it_xch.as<double>(); a.sync = data->it_sync.as<unsigned>(); a.vgrid = resident_grid;
        a.n_global = (double)data->n_global; a.refine_limit = refine_limit; a.atol = test.atol; a.rtol = test.rtol;
        a.ll_offset = (double)d * log_two_pi() / 2;                       // (read(): the same expression)
        a.max_steps = max_steps; a.history = history; a.result = result;
        // MLHIP_RESIDENT_PROFILE=1: per-phase clock stamps of workgroup 0, averaged over the iterations, on stderr (diagnostic)
        static const bool profile = [] { const char* e = std::getenv("MLHIP_RESIDENT_PROFILE"); return e && e[0] == '1'; }();
        DevBuf stamps;
        if (profile) {
            stamps.reserve(sizeof(unsigned long long) * mstats::kResidentStamps * max_steps);
            HIP_CHECK(hipMemsetAsync(stamps.p, 0, stamps.bytes, ctx->stream));
            a.profile = stamps.as<unsigned long long>();
        }
        struct Release { DevBuf& b; ~Release() { b.release(); } } release_stamps{stamps};
        bool ok = false;
        ctx->timed("em_resident", [&] { ok = mstats::launch_em_resident(a, ctx->stream); });
        if (!ok) throw std::runtime_error("resident EM kernel not instantiated for this shape");
        HIP_CHECK(hipGetLastError());
        ctx->sync();
        const uint32_t status = result[0], evaluated = result[1];
        if (profile && evaluated > 1 && evaluated <= max_steps) {
            constexpr int NS = mstats::kResidentStamps;
            std::vector<unsigned long long> t((size_t)NS * evaluated);
            HIP_CHECK(hipMemcpy(t.data(), stamps.p, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost));
            // (from, to) stamp slots of the named spans; slot 0 of the NEXT iteration closes the last one
            struct Span { const char* name; int from, to; };
            const Span spans[] = {{"pass", 0, 1}, {"  densities", 0, 8}, {"  exp+log+1/s", 8, 9}, {"  statistics", 9, 10}, {"  lane fold+barrier", 11, 1},
                                  {"publish", 1, 2}, {"exchange (wait + gather)", 2, 4}, {"sums", 4, 5}, {"closing", 5, 6},
                                  {"  statistics->cov", 5, 12}, {"  factorization", 12, 13}, {"  inverse", 13, 14}, {"  store L, W", 14, 15},
                                  {"  logarithms", 15, 16}, {"  flags, log det, c", 16, 17}, {"  records+barrier", 17, 6}, {"outputs", 6, 7}};
            std::string line;
            char buf[96];
            const double us = 0.01 / (evaluated - 1);                     // 100 MHz stamps; iteration 0 (cold loads) left out
            for (const Span& sp : spans) {
                double sum = 0;
                for (uint32_t i = 1; i < evaluated; ++i) sum += (double)(long long)(t[(size_t)NS * i + sp.to] - t[(size_t)NS * i + sp.from]);
                std::snprintf(buf, sizeof buf, "%s %.2f, ", sp.name, sum * us);
                line += buf;
            }
            double whole = 0;
            for (uint32_t i = 1; i < evaluated; ++i) whole += (double)(t[(size_t)NS * i] - t[(size_t)NS * (i - 1)]);
            std::fprintf(stderr, "[mlhip] resident loop, N=%u d=%d K=%d grid=%d, us per iteration (thread 0 of workgroup 0): %siteration %.2f\n",
                         data->n, d, K, resident_grid, line.c_str(), whole * us);
        }
        if (status == 3 || status == 0 || evaluated > max_steps) {          // a wait gave up (or nothing came back): the ordinary loop, from the start
            *resume_at = 0;
            launched = 0;
            resident_gave_up = true;                                      // (the ring may hold later iterations' records by now)
            return false;
        }
        data->n_ll = resident_grid;                                       // (what launch_fused_step leaves behind on the host side)
        data->have_estep = true;
        data->lw_valid = false;
        data->stats_mode = kFromLogResp;
        data->stats_resp = data->lw.as<double>();
        data->stats_ld = data->ldr;
        for (uint32_t i = 0; i < evaluated; ++i) {                        // the host's own test over the history: the same decisions
            const bool stopped = test(i, history[i]);
            if (stopped != (status == 1 && i + 1 == evaluated && result[2] != 0))
                throw std::runtime_error("resident EM loop: the device's convergence test disagrees with the host's");
        }
        launched = evaluated;
        if (status == 2) {                                                // iteration `evaluated` flagged: run it again, closed on the host
            if (evaluated > 0) fetch_parameters((int)(evaluated % 3));
            *resume_at = evaluated;
            return false;
        }
        fetch_parameters((int)(evaluated % 3));                           // P_(last + 1): the newest parameters
        finish(evaluated - 1);
        return true;
    }

    struct Verdict {
        double ll;
        bool flagged;        // some component needs the refinement pass (far, tight cluster)
        double cmax;         // max |W (mu - shift)| over the new records: the FOLD criterion of the NEXT E-step
    };
    /// What iteration i reported (its info block must have arrived: event / stream waited for).
    Verdict read(uint32_t i)
    {
        const int out = (int)((i + 1) % 3);
        const double* inf = host_info(out);
        Verdict v{inf[0] / (double)data->n_global - (double)d * log_two_pi() / 2, false, 0.0};   // ML/EM.cpp:197-198, 211
        for (int k = 0; k < K; ++k) {
            v.flagged = v.flagged || inf[1 + k] != 0.0;
            v.cmax = std::max(v.cmax, inf[1 + K + k]);
        }
        if (diag) shadow[out].assign(inf + n_info, inf + n_pack);
        return v;
    }

    /// The records of ring slot `slot` become data->params_dev -- what the per-step functions (refinement pass, ensure_lw) read.
    void bring_to_params_dev(int slot)
    {
        if (rec[slot] == &data->params_dev) return;
        int j = 0;
        while (rec[j] != &data->params_dev) ++j;
        std::swap(*rec[slot], data->params_dev);            // the contents change places ...
        std::swap(rec[slot], rec[j]);                       // ... and the ring's names follow them
        data->estep_fold = fold[slot];
    }

    /// Parameters P_slot -> the caller's arrays.
    void fetch_parameters(int slot)
    {
        HIP_CHECK(hipMemcpyAsync(mixing, pack_mixing(slot), sizeof(double) * K, hipMemcpyDefault, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(means, pack_means(slot), sizeof(double) * K * d, hipMemcpyDefault, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(covs, pack_covs(slot), sizeof(double) * n_cov, hipMemcpyDefault, ctx->stream));
        ctx->sync();
    }

    /// Iteration i (evaluated, nothing behind it on the stream) closed on the HOST with its refinement pass -- the arithmetic of
    /// mlhip_em_step; the new parameters land in the caller's arrays, the new records in ring slot (i + 1) % 3.
    void close_on_host(uint32_t i)
    {
        const int in = (int)(i % 3), out = (int)((i + 1) % 3);
        bring_to_params_dev(in);
        HIP_CHECK(hipMemcpyAsync(data->stats_host.p, data->stats_dev.p, sizeof(double) * ((size_t)K * F + 1), hipMemcpyDeviceToHost,
                                 ctx->stream));
        ctx->sync();
        if (diag) {
            host::finalize_mstep_diag(d, K, data->stats_host.as<double>(), data->shift.data(), (double)data->n_global, mixing, means, covs);
            refine_diag(data, K, mixing, means, covs);
            upload_diag_records(data, K, mixing, means, covs, *rec[out]);
            shadow[out].assign(mixing, mixing + K);
            shadow[out].insert(shadow[out].end(), means, means + (size_t)K * d);
            shadow[out].insert(shadow[out].end(), covs, covs + n_cov);
        } else {
            finalize_out(data, K, mixing, means, covs);
            const int variant = data->estep_variant;
            prepare_estep(data, K, mixing, means, covs, rec[out]);
            fold[out] = data->estep_fold;
            data->estep_fold = fold[in];                     // (still describes the records in params_dev)
            if (data->estep_variant != variant) throw std::runtime_error("E-step record layout changed inside a fit");
        }
    }

    /// Device state as the per-step entry points leave it: the records of the LAST evaluated E-step in params_dev; whatever a
    /// speculative iteration overwrote (log-responsibilities, lse) is rebuilt from them on demand.
    void finish(uint32_t last)
    {
        bring_to_params_dev((int)(last % 3));
        data->estep_fold = fold[last % 3];
        if (launched > last + 1) data->lw_valid = false;
        if (diag) {
            const std::vector<double>& sh = shadow[last % 3];
            data->diag_mixing.assign(sh.begin(), sh.begin() + K);
            data->diag_means.assign(sh.begin() + K, sh.begin() + K + (size_t)K * d);
            data->diag_vars.assign(sh.begin() + K + (size_t)K * d, sh.end());
        }
    }

    /// Whether iteration i + 1 may be launched before the host has seen iteration i. Only where an iteration is short enough for the
    /// host's share to matter -- the speculative iteration is thrown away once per fit, which a long iteration never earns back
    /// (N = 10M, d = 8, K = 32: 2.2 ms per iteration against ~10 us saved) -- and never with a host-side all-reduce. Decided from the
    /// GLOBAL row count: every rank (every shard of a device group) must take the same loop, or a lagged rank's speculative
    /// all-reduce would meet another rank's end-of-fit exchange. MLHIP_LAGGED=0: off; MLHIP_LAGGED_WORK: the bound on
    /// N K d^2 per rank (default 2e9; diagonal: N K d <= 1e9).
    bool lagged_applies(uint32_t max_steps) const
    {
        static const double work_limit = [] { const char* e = ab_env("MLHIP_LAGGED_WORK"); return e ? std::atof(e) : 2.0e9; }();
        const double pair_work = (double)data->n_global / (double)std::max(1, ctx->world_size) * K * (diag ? d : d * d);
        return !(ab_env("MLHIP_LAGGED") && ab_env("MLHIP_LAGGED")[0] == '0') && (!ctx->reduce_fn || ctx->reduce_on_device) && max_steps >= 2 &&
               pair_work <= (diag ? 1.0e9 : work_limit);
    }

    /// Lagged policy: the convergence test of ML/EM.cpp:161-168 fires one iteration late and the speculative iteration is simply
    /// dropped -- the ring keeps the inputs and outputs of iteration i intact while i + 1 runs, so the results are bit-identical to
    /// the synchronous loop. The matrix-core E-step runs in its EXACT form here (FOLD is a per-iteration decision of the host; the
    /// records carry both vectors). Returns true when the loop is over; false when a refinement flag at iteration *resume_at hands
    /// over to the synchronous policy (which re-runs that iteration and closes it on the host).
    bool run_lagged(ConvergenceTest& test, uint32_t max_steps, uint32_t* resume_at)
    {
        fold[0] = fold[1] = fold[2] = false;
        launch(0);
        for (uint32_t i = 0; i < max_steps; ++i) {
            if (i + 1 < max_steps) launch(i + 1);
            HIP_CHECK(hipEventSynchronize(data->it_event[(i + 1) % 3]));
            const Verdict v = read(i);
            if (v.flagged) {
                ctx->sync();                                 // (the speculative iteration, if any: let it finish, then forget it)
                if (i > 0) fetch_parameters((int)(i % 3));   // P_i, the inputs of the iteration that is run again
                launched = i;
                *resume_at = i;
                return false;
            }
            if (test(i, v.ll) || i + 1 == max_steps) {
                ctx->sync();
                fetch_parameters((int)((i + 1) % 3));        // P_(i+1): the newest parameters
                finish(i);
                return true;
            }
        }
        return true;   // (not reached: max_steps >= 2)
    }

    /// Synchronous policy, from iteration `first` on: launch, wait, decide. The host picks the E-step form of the next iteration
    /// (FOLD while every |W (mu - shift)| is small) and closes a flagged iteration itself.
    void run_synchronous(ConvergenceTest& test, uint32_t first, uint32_t max_steps)
    {
        bool latest_on_host = true;
        uint32_t last = first;
        for (uint32_t i = first; i < max_steps; ++i) {
            PhaseTrace tr;
            launch(i);
            ctx->sync();
            tr.mark("iteration (device close)");
            const Verdict v = read(i);
            if (v.flagged) {
                close_on_host(i);
                latest_on_host = true;
            } else {
                latest_on_host = false;
                fold[(i + 1) % 3] = fold_allowed && data->estep_variant == 2 && data->D <= kRegDim && v.cmax <= kEstepFoldLimit;
            }
            last = i;
            if (test(i, v.ll)) break;
        }
        if (!latest_on_host) fetch_parameters((int)((last + 1) % 3));
        finish(last);
    }
};

/// d > 1024 (no device closing), MLHIP_DEVICE_CLOSE=0, experimental record layouts: the loop over the per-step functions.
void host_closing_loop(mlhip_data* data, int K, bool diag, double* mixing, double* means, double* covs, uint32_t max_steps,
                       ConvergenceTest& test)
{
    for (uint32_t step = 0; step < max_steps; ++step) {
        double ll = 0;
        if (diag) em_step_diag(data, K, mixing, means, covs, &ll, mixing, means, covs);
        else em_step_full(data, K, mixing, means, covs, &ll, mixing, means, covs);
        if (test(step, ll)) break;
    }
}

}  // namespace

void em_iterate(mlhip_data* data, int K, bool diag, double* mixing, double* means, double* covs, uint32_t max_steps, double atol,
                double rtol, uint32_t* steps_done, int* converged, double* log_likelihood, double* history)
{
    *steps_done = 0;
    *converged = 0;
    ConvergenceTest test{atol, rtol, steps_done, converged, log_likelihood, history};
    EmLoop loop(data, K, diag, mixing, means, covs);
    if (!loop.prepare()) {
        host_closing_loop(data, K, diag, mixing, means, covs, max_steps, test);
        return;
    }
    uint32_t first = 0;
    if (loop.resident_grid > 0) {
        if (loop.run_resident(test, max_steps, &first)) return;
        if (loop.resident_gave_up) loop.prepare();                        // records R_0 again, from the caller's (untouched) arrays
        loop.run_synchronous(test, first, max_steps);
        return;
    }
    if (loop.lagged_applies(max_steps) && loop.run_lagged(test, max_steps, &first)) return;
    loop.run_synchronous(test, first, max_steps);
}

}  // namespace mlhip_rt
