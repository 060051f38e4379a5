// K-means family of the C ABI: assignment + exact update sums per step, the step loop of KMeans::fit_once in one call.
#include <cmath>
#include <vector>

#include "internal.hpp"

namespace mlhip_rt {


void ensure_km_workspace(mlhip_data* dt, int K)
{
    mlhip_ctx* ctx = dt->ctx;
    for (int b = 0; b < 2; ++b) dt->km_labels[b].reserve(sizeof(uint32_t) * dt->n_pad);
    dt->km_mind.reserve(sizeof(double) * dt->n_pad);
    if (!dt->km_scale.p) {
        // Per-dimension power-of-two scale of the exact fixed-point sums: |x_j| * scale_j < 2^94 (device/kmeans.hip).
        DevBuf scratch, mx;
        scratch.reserve(sizeof(double) * 1024 * dt->d);
        mx.reserve(sizeof(double) * dt->d);
        launch_column_maxabs(dt->xt.as<double>(), dt->ldx, dt->d, dt->n, scratch.as<double>(), mx.as<double>(), ctx->stream);
        std::vector<double> m(dt->d);
        HIP_CHECK(hipMemcpyAsync(m.data(), mx.p, sizeof(double) * dt->d, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        scratch.release(); mx.release();
        // Every rank must cut its coordinates on the SAME fixed-point grid (the limb sums are added across ranks): the
        // column maxima are exchanged through the sum hook, one slot per rank, and every rank takes the maximum.
        if (ctx->world_size > 1) {
            std::vector<double> all((size_t)ctx->world_size * dt->d, 0.0);
            for (int j = 0; j < dt->d; ++j) all[(size_t)ctx->rank * dt->d + j] = m[j];
            ctx->allreduce_host(all.data(), all.size());
            for (int r = 0; r < ctx->world_size; ++r)
                for (int j = 0; j < dt->d; ++j) {
                    const double v = all[(size_t)r * dt->d + j];
                    if (!(v <= m[j])) m[j] = v;                            // (keeps a NaN / inf of any rank)
                }
        }
        for (int j = 0; j < dt->d; ++j) {
            if (!std::isfinite(m[j]))
                throw DomainError("K-means: the data contain non-finite values (the exact fixed-point update sums need finite coordinates)");
            int e = 0;
            if (m[j] > 0) (void)std::frexp(m[j], &e);   // m < 2^e
            m[j] = std::ldexp(1.0, 94 - e);
        }
        dt->km_scale.reserve(sizeof(double) * dt->d);
        HIP_CHECK(hipMemcpyAsync(dt->km_scale.p, m.data(), sizeof(double) * dt->d, hipMemcpyHostToDevice, ctx->stream));
        ctx->sync();
    }
    const int Dp = (dt->D + 3) & ~3;                              // (the K-means kernels may run on a zero-padded copy)
    dt->km_cent.reserve(sizeof(double) * (size_t)K * Dp);
    dt->km_cnorm.reserve(sizeof(double) * (size_t)((K + 15) & ~15));
    dt->km_partials.reserve(sizeof(double) * kmeans_scratch_doubles(dt->d, K, ctx->num_cus));
    const size_t ob = sizeof(double) * (2 + (size_t)K * (dt->d + 1));
    dt->km_out.reserve(ob);
    const size_t hb = ob > sizeof(double) * (size_t)K * Dp ? ob : sizeof(double) * (size_t)K * Dp;
    dt->km_host.reserve(hb);
}


KmBlock km_block(mlhip_data* dt, int K)
{
    mlhip_ctx* ctx = dt->ctx;
    ensure_km_workspace(dt, K);
    // The matrix-core kernel needs a multiple of 4 dimensions. For d = 1, 2, 3, 5, 6 (stored with D = d or 6 rows) and many
    // clusters it still beats the direct-form kernel (d = 6, K = 256: 1.9 -> 1.2 ms at N = 10M), so such blocks get a copy
    // padded with zero rows once: zero coordinates add exactly 0 to every distance, labels and sums are unchanged.
    KmBlock b{dt->xt.as<double>(), dt->D};
    if (b.D % 4 != 0 && K >= 128 && !std::getenv("MLHIP_KMEANS")) {
        const int Dp = (b.D + 3) & ~3;
        if (!dt->km_xt_pad.p) {
            dt->km_xt_pad.reserve(sizeof(double) * dt->ldx * Dp);
            HIP_CHECK(hipMemsetAsync(dt->km_xt_pad.p, 0, sizeof(double) * dt->ldx * Dp, ctx->stream));
            HIP_CHECK(hipMemcpyAsync(dt->km_xt_pad.p, dt->xt.p, sizeof(double) * dt->ldx * b.D, hipMemcpyDeviceToDevice, ctx->stream));
        }
        b.D = Dp;
        b.xt = dt->km_xt_pad.as<double>();
    }
    return b;
}


/// Host centroids [K][d] -> the device table km_cent [K][D] (padded coordinates zero).
void km_upload_centroids(mlhip_data* dt, int K, const KmBlock& b, const double* centroids)
{
    mlhip_ctx* ctx = dt->ctx;
    double* ch = dt->km_host.as<double>();
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < b.D; ++j) ch[(size_t)k * b.D + j] = j < dt->d ? centroids[(size_t)k * dt->d + j] : 0.0;
    HIP_CHECK(hipMemcpyAsync(dt->km_cent.p, ch, sizeof(double) * (size_t)K * b.D, hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();   // km_host is reused for the results
}


/// Assignment (+ optional accumulation) against the table in km_cent, partials reduced into km_out =
/// [inertia, changed, counts, sums] and summed across ranks there when the all-reduce works on device memory.
bool km_launch(mlhip_data* dt, int K, const KmBlock& b, bool accumulate, double* min_dist_out, double* close_next, double* close_mirror)
{
    mlhip_ctx* ctx = dt->ctx;
    const int nxt = dt->km_cur ^ 1;
    KmeansArgs a{};
    a.xt = b.xt; a.ldx = dt->ldx; a.n = dt->n; a.D = b.D; a.d = dt->d;
    a.centroids = dt->km_cent.as<double>(); a.K = K;
    a.scale = dt->km_scale.as<double>();
    a.labels = dt->km_labels[nxt].as<uint32_t>();
    a.old_labels = dt->km_labels[dt->km_cur].as<uint32_t>();
    a.have_old = dt->km_have_old ? 1 : 0;
    a.min_dist = min_dist_out ? min_dist_out : dt->km_mind.as<double>();   // a distance-only probe writes elsewhere
    a.accumulate = accumulate ? 1 : 0;
    a.partials = dt->km_partials.as<double>(); a.partials_capacity = dt->km_partials.bytes / sizeof(double);
    a.cnorm = dt->km_cnorm.as<double>();
    a.out = dt->km_out.as<double>();
    int rc = 0;
    ctx->timed("kmeans_assign", [&] { rc = launch_kmeans_assign(a, ctx->num_cus, ctx->stream); });
    if (rc == -1) throw Unsupported("K-means kernel not instantiated for this dimension");
    if (rc <= 0) throw std::runtime_error("K-means kernel launch failed");
    bool fused = false;
#ifdef MLHIP_EXPERIMENTS
    // Single rank, the step loop: reduction and closing arithmetic in ONE launch -- measured slower than the two (kmeans.hip
    // kmeans_reduce_close_kernel); MLHIP_KMEANS_FUSED=1 in a `make EXPERIMENTS=1` library selects it (A/B runs)
    static const bool fuse_wanted = [] { const char* e = std::getenv("MLHIP_KMEANS_FUSED"); return e && e[0] == '1'; }();
    fused = close_next && accumulate && !ctx->reduce_fn && fuse_wanted;
    if (fused) {
        if (!dt->km_ticket.p) {
            dt->km_ticket.reserve(64);
            HIP_CHECK(hipMemsetAsync(dt->km_ticket.p, 0, 64, ctx->stream));
            dt->km_ticket_base = 0;
        }
        dt->km_ticket_base += launch_kmeans_reduce_close(a, rc, b.D, close_next, close_mirror, dt->km_ticket.as<unsigned>(), dt->km_ticket_base, ctx->stream);
    }
#else
    (void)close_next; (void)close_mirror;
#endif
    if (!fused) launch_kmeans_reduce(a, rc, ctx->stream);
    HIP_CHECK(hipGetLastError());
    dt->km_cur = nxt;
    dt->km_have_old = true;
    if (ctx->reduce_fn && ctx->reduce_on_device) {
        const size_t count = 2 + (accumulate ? (size_t)K * (dt->d + 1) : 0);
        ctx->reduce_device(dt->km_out.as<double>(), count);
    }
    return fused;
}


/// km_out -> km_host (`count` doubles), summed across ranks on the host when the all-reduce works on host memory.
void km_fetch(mlhip_data* dt, size_t count)
{
    mlhip_ctx* ctx = dt->ctx;
    double* ch = dt->km_host.as<double>();
    HIP_CHECK(hipMemcpyAsync(ch, dt->km_out.p, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->reduce_fn && !ctx->reduce_on_device) {
        if (ctx->reduce_fn(ctx->reduce_user, ch, count, 0, ctx->stream) != 0) throw hook_failure();
    }
}


/// Assignment (+ optional accumulation); leaves all-reduced [inertia, changed, counts, sums] in km_host.
void run_kmeans(mlhip_data* dt, int K, const double* centroids, bool accumulate, double* min_dist_out)
{
    const KmBlock b = km_block(dt, K);
    km_upload_centroids(dt, K, b, centroids);
    km_launch(dt, K, b, accumulate, min_dist_out);
    km_fetch(dt, 2 + (accumulate ? (size_t)K * (dt->d + 1) : 0));
}


/// update_step's closing arithmetic on the host (ML/KMeans.cpp:180-192 as sums / counts; empty cluster -> origin, :184).
void km_close_host(const double* r, int K, int d, double* counts, double* centroids_out)
{
    for (int k = 0; k < K; ++k) {
        const double c = r[2 + k];
        if (counts) counts[k] = c;
        for (int j = 0; j < d; ++j) centroids_out[(size_t)k * d + j] = c > 0 ? r[2 + K + (size_t)k * d + j] / c : 0.0;
    }
}


/// The step loop of KMeans::fit_once (ML/KMeans.cpp:80-110). With the all-reduce on device memory (or none) the centroid
/// table never leaves the device between trips: sums -> means -> next table by launch_kmeans_close, one read-back per trip
/// for the two stopping tests. With a host-memory all-reduce (gloo rehearsals) every trip goes through run_kmeans.
void km_iterate(mlhip_data* dt, int K, double* centroids, double* old_centroids, uint32_t max_steps, double atol,
                uint32_t* steps_done, int* converged, double* inertia, double* counts)
{
    mlhip_ctx* ctx = dt->ctx;
    const int d = dt->d;
    const size_t kd = (size_t)K * d;
    const bool device_route = !(ctx->reduce_fn && !ctx->reduce_on_device) && !ab_env("MLHIP_KMEANS_HOST_LOOP");
    const KmBlock b = km_block(dt, K);
    std::vector<double> cur(centroids, centroids + kd), old(kd, 0.0), upd(kd);
    // Small blocks with few clusters in the dimensions of the direct-form kernel: the whole loop in ONE launch of one workgroup
    // (device/kmeans_resident.hip; bit-identical to the launches below; MLHIP_RESIDENT=0: off).
    const char* const res_env = std::getenv("MLHIP_RESIDENT");                 // (per call, as the EM loop reads it: the tests flip it)
    const bool resident_allowed = !(res_env && res_env[0] == '0');
    if (device_route && !ctx->reduce_fn && ctx->world_size <= 1 && b.xt == dt->xt.as<double>() && resident_allowed &&
        kmeans_resident_supported(b.D, d, K, dt->n)) {
        const size_t n_out = 4 + (size_t)K + 2 * kd;
        dt->km_host.reserve(sizeof(double) * (n_out > (size_t)K * b.D ? n_out : (size_t)K * b.D));
        km_upload_centroids(dt, K, b, cur.data());
        KmResidentArgs a{};
        a.xt = b.xt; a.ldx = dt->ldx; a.n = dt->n; a.D = b.D; a.d = d; a.K = K;
        a.cent = dt->km_cent.as<double>(); a.scale = dt->km_scale.as<double>();
        a.labels[0] = dt->km_labels[0].as<uint32_t>(); a.labels[1] = dt->km_labels[1].as<uint32_t>();
        a.label_buf = dt->km_cur; a.have_old = dt->km_have_old ? 1 : 0;
        a.min_dist = dt->km_mind.as<double>();
        a.max_steps = max_steps; a.atol = atol;
        a.out = dt->km_host.as<double>();
        bool ok = false;
        ctx->timed("kmeans_resident", [&] { ok = launch_kmeans_resident(a, ctx->stream); });
        if (!ok) throw std::runtime_error("resident K-means kernel not instantiated for this dimension");
        HIP_CHECK(hipGetLastError());
        ctx->sync();
        const double* r = dt->km_host.as<double>();
        *steps_done = (uint32_t)std::llround(r[0]);
        *converged = r[1] != 0.0 ? 1 : 0;
        *inertia = r[2];
        dt->km_cur = (int)std::llround(r[3]);
        dt->km_have_old = true;
        if (counts) std::copy(r + 4, r + 4 + K, counts);
        std::copy(r + 4 + K, r + 4 + K + kd, centroids);
        if (old_centroids) std::copy(r + 4 + K + kd, r + 4 + K + 2 * kd, old_centroids);
        return;
    }
    if (device_route) {
        dt->km_cent_next.reserve(sizeof(double) * (size_t)K * b.D);
        km_upload_centroids(dt, K, b, cur.data());
    }
    *converged = 0;
    *steps_done = 0;
    for (uint32_t step = 0; step < max_steps; ++step) {
        if (device_route) {
            // (the closing arithmetic writes the block into the pinned km_host as well: no copy-engine transfer in the loop)
            static const bool mirror = [] { const char* e = ab_env("MLHIP_KMEANS_MIRROR"); return !(e && e[0] == '0'); }();
            double* const pinned = mirror ? dt->km_host.as<double>() : nullptr;
            if (!km_launch(dt, K, b, true, nullptr, dt->km_cent_next.as<double>(), pinned))
                launch_kmeans_close(dt->km_out.as<double>(), K, d, b.D, dt->km_cent_next.as<double>(), pinned, ctx->stream);
            HIP_CHECK(hipGetLastError());
            if (mirror) ctx->sync(); else km_fetch(dt, 2 + (size_t)K * (d + 1));
            const double* r = dt->km_host.as<double>();
            if (counts) std::copy(r + 2, r + 2 + K, counts);
            std::copy(r + 2 + K, r + 2 + K + kd, upd.begin());
        } else {
            run_kmeans(dt, K, cur.data(), true);
            km_close_host(dt->km_host.as<double>(), K, d, counts, upd.data());
        }
        const double* r = dt->km_host.as<double>();
        *inertia = r[0];
        const uint64_t changed = (uint64_t)std::llround(r[1]);
        ++*steps_done;
        if (step > 0 && changed == 0) {   // same labels twice (:84-89): the centroids stay as they are
            *converged = 1;
            break;
        }
        old.swap(cur);                    // update_step (:180-192)
        cur.swap(upd);
        if (device_route) std::swap(dt->km_cent, dt->km_cent_next);
        if (step > 0) {
            double shift = 0;
            for (size_t t = 0; t < kd; ++t) {
                const double delta = cur[t] - old[t];
                shift += delta * delta;
            }
            if (shift < atol) {           // (:103-108) one more assignment under the final centroids
                if (device_route) {
                    km_launch(dt, K, b, false, nullptr);
                    km_fetch(dt, 2);
                } else {
                    run_kmeans(dt, K, cur.data(), false);
                }
                *inertia = dt->km_host.as<double>()[0];
                *converged = 1;
                break;
            }
        }
    }
    std::copy(cur.begin(), cur.end(), centroids);
    if (old_centroids) std::copy(old.begin(), old.end(), old_centroids);
}

}  // namespace mlhip_rt

extern "C" {


int mlhip_kmeans_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* inertia,
                      uint64_t* n_changed, double* counts, double* centroids_out)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::kmeans_step(ctx, data, K, true, centroids, inertia, n_changed, counts, centroids_out); return; }
        check_em_args(ctx, data, K);
        require(centroids && inertia && n_changed && counts && centroids_out, "null argument");
        run_kmeans(data, (int)K, centroids, true);
        const double* r = data->km_host.as<double>();
        *inertia = r[0];
        *n_changed = (uint64_t)std::llround(r[1]);
        km_close_host(r, (int)K, data->d, counts, centroids_out);
    });
}

int mlhip_kmeans_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* centroids, double* old_centroids,
                         uint32_t max_steps, double absolute_tolerance, uint32_t* steps_done, int* converged,
                         double* inertia, double* counts)
{
    return guarded([&] {
        if (ctx && ctx->group) {
            require(max_steps >= 1, "at least one step");
            require(absolute_tolerance >= 0, "negative tolerance");
            grp::kmeans_iterate(ctx, data, K, centroids, old_centroids, max_steps, absolute_tolerance, steps_done, converged, inertia, counts);
            return;
        }
        check_em_args(ctx, data, K);
        require(centroids && steps_done && converged && inertia, "null argument");
        require(max_steps >= 1, "at least one step");
        require(absolute_tolerance >= 0, "negative tolerance");
        km_iterate(data, (int)K, centroids, old_centroids, max_steps, absolute_tolerance, steps_done, converged, inertia, counts);
        ctx->check_ranks_agree("the K-means centroids", {{centroids, (size_t)K * data->d}, {inertia, 1}});
    });
}

int mlhip_kmeans_assign(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* inertia,
                        uint64_t* n_changed)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::kmeans_step(ctx, data, K, false, centroids, inertia, n_changed, nullptr, nullptr); return; }
        check_em_args(ctx, data, K);
        require(centroids && inertia && n_changed, "null argument");
        run_kmeans(data, (int)K, centroids, false);
        const double* r = data->km_host.as<double>();
        *inertia = r[0];
        *n_changed = (uint64_t)std::llround(r[1]);
    });
}

int mlhip_kmeans_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t* labels)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::kmeans_labels(ctx, data, labels); return; }
        check_em_args(ctx, data, 1);
        require(labels || data->n == 0, "null argument");
        require(data->km_have_old, "no K-means assignment on the device yet");
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(labels), 0, data->km_labels[data->km_cur].as<char>(), 0,
                         sizeof(uint32_t) * data->n, 1);
    });
}

int mlhip_kmeans_distances(mlhip_ctx* ctx, mlhip_data* data, double* dist2)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::kmeans_distances(ctx, data, dist2); return; }
        check_em_args(ctx, data, 1);
        require(dist2 || data->n == 0, "null argument");
        require(data->km_have_old, "no K-means assignment on the device yet");
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(dist2), 0, data->km_mind.as<char>(), 0, sizeof(double) * data->n, 1);
    });
}

int mlhip_kpp_draw(mlhip_ctx* ctx, mlhip_data* data, const double* centroid, int first, double u, uint64_t first_row, uint64_t* index,
                   int* certain, double* weights_out)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::kpp_draw(ctx, data, centroid, first, u, first_row, index, certain, weights_out); return; }
        check_em_args(ctx, data, 1);
        require(centroid && index && certain, "null argument");
        require(data->n_global >= 2, "at least two rows");
        require(u >= 0.0 && u < 1.0, "u must be a canonical uniform draw");
        const uint64_t n_global = data->n_global;
        require(first_row + data->n <= n_global, "first_row beyond the sample");
        // distances to the new centroid -> km_probe (as mlhip_min_squared_distances, label history untouched)
        const int cur = data->km_cur;
        const bool have = data->km_have_old;
        data->km_probe.reserve(sizeof(double) * data->n_pad);
        const KmBlock b = km_block(data, 1);
        km_upload_centroids(data, 1, b, centroid);
        km_launch(data, 1, b, false, data->km_probe.as<double>());
        if (have) data->km_cur = cur;
        data->km_have_old = have;
        const int nb = kpp_blocks(data->n);
        data->kpp_w.reserve(sizeof(double) * data->n_pad);
        data->kpp_scr.reserve(sizeof(double) * (2 * (size_t)nb + 4));
        double* bsum = data->kpp_scr.as<double>();
        double* boff = bsum + nb;
        double* out = boff + nb;
        // |cp_i - c~_i| <= (4 N + 16384) 2^-53 (data_kernels.hip); MLHIP_KPP_DELTA_SCALE widens it (tests: forces the host path)
        static const double scale = [] { const char* e = std::getenv("MLHIP_KPP_DELTA_SCALE"); return e ? std::atof(e) : 1.0; }();
        const double delta = scale * (4.0 * (double)n_global + 16384.0) * 0x1p-53;
        double* res = data->km_host.as<double>();
        ctx->timed("kpp_draw", [&] {
            launch_kpp_update(data->kpp_w.as<double>(), data->km_probe.as<double>(), data->n, first ? 1 : 0, (double)(n_global - 1), bsum,
                              boff, out, ctx->stream);
        });
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(res, out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        // the ranks' sums in rank order: this rank's offset and the total are the same sequence of additions on every rank
        int world = 1, rank = 0;
        if (ctx->reduce_fn) { world = ctx->world_size; rank = ctx->rank; }
        std::vector<double> sums((size_t)world, 0.0);
        sums[(size_t)rank] = res[0];
        if (world > 1) ctx->allreduce_host(sums.data(), sums.size());
        double offset = 0.0, total = 0.0;
        for (int r = 0; r < world; ++r) {
            if (r == rank) offset = total;
            total += sums[(size_t)r];
        }
        ctx->timed("kpp_draw", [&] {
            launch_kpp_find(data->kpp_w.as<double>(), data->n, bsum, boff, offset, total, u, delta, first_row, n_global, out, ctx->stream);
        });
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(res, out, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        double lo = res[1], hi = res[2];
        if (world > 1) {
            std::vector<double> cand(2 * (size_t)world, 0.0);
            cand[2 * (size_t)rank] = lo;
            cand[2 * (size_t)rank + 1] = hi;
            ctx->allreduce_host(cand.data(), cand.size());
            for (int r = 0; r < world; ++r) {
                lo = std::min(lo, cand[2 * (size_t)r]);
                hi = std::min(hi, cand[2 * (size_t)r + 1]);
            }
        }
        const bool ok = std::isfinite(total) && total > 0.0 && lo == hi;
        *certain = ok ? 1 : 0;
        *index = ok ? (uint64_t)lo : 0;
        if (!ok && weights_out && data->n)
            download_columns(ctx, reinterpret_cast<char*>(weights_out), 0, data->kpp_w.as<char>(), 0, sizeof(double) * data->n, 1);
    });
}

int mlhip_kpp_weights(mlhip_ctx* ctx, mlhip_data* data, double* weights_out)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::kpp_weights(ctx, data, weights_out); return; }
        check_em_args(ctx, data, 1);
        require(weights_out || data->n == 0, "null argument");
        require(data->kpp_w.p != nullptr, "no K-means++ draw on the device yet");
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(weights_out), 0, data->kpp_w.as<char>(), 0, sizeof(double) * data->n, 1);
    });
}

int mlhip_min_squared_distances(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* dist2)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::min_squared_distances(ctx, data, K, centroids, dist2); return; }
        check_em_args(ctx, data, K);
        require(centroids && (dist2 || data->n == 0), "null argument");
        // Must disturb neither the label history used for n_changed nor the per-sample distances of the last assignment
        // (mlhip_kmeans_distances): the labels go to the spare buffer, the distances to a buffer of their own.
        const int cur = data->km_cur;
        const bool have = data->km_have_old;
        data->km_probe.reserve(sizeof(double) * data->n_pad);
        run_kmeans(data, (int)K, centroids, false, data->km_probe.as<double>());
        if (have) {
            // The assignment wrote labels into the *other* buffer; keep the previous labels current.
            data->km_cur = cur;
        }
        data->km_have_old = have;
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(dist2), 0, data->km_probe.as<char>(), 0, sizeof(double) * data->n, 1);
    });
}

}  // extern "C"
