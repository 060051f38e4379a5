// Data family of the C ABI: upload of the caller's sample block into the dimension-major HBM layout (pinned, double-buffered
// staging), the shift, downloads; and the one-pass helpers on the resident block (sample covariance, X X^T / X y, RandomPartition).
#include "internal.hpp"

namespace mlhip_rt {


/// memcpy split over a few threads: one core moves ~10 GB/s (less into untouched pages), below the PCIe rate it feeds.
void copy_bytes(void* dst, const void* src, size_t bytes)
{
    constexpr size_t kPerThread = size_t(8) << 20;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t parts = std::min<size_t>(std::min<unsigned>(hw, 4u), bytes / kPerThread);
    if (parts < 2) { std::memcpy(dst, src, bytes); return; }
    const size_t each = (bytes / parts + 4095) & ~size_t(4095);
    std::vector<std::thread> pool;
    for (size_t t = 1; t < parts; ++t) {
        const size_t off = t * each;
        if (off >= bytes) break;
        pool.emplace_back([=] { std::memcpy((char*)dst + off, (const char*)src + off, std::min(each, bytes - off)); });
    }
    std::memcpy(dst, src, std::min(each, bytes));
    for (auto& th : pool) th.join();
}


void finish_upload(mlhip_data* dt)
{
    mlhip_ctx* ctx = dt->ctx;
    // shift = global column mean (all-reduced sums and counts).
    DevBuf scratch, sums;
    scratch.reserve(sizeof(double) * 1024 * dt->d);
    sums.reserve(sizeof(double) * dt->d);
    launch_column_sums(dt->xt.as<double>(), dt->ldx, dt->d, dt->n, scratch.as<double>(), sums.as<double>(), ctx->stream);
    std::vector<double> v(dt->d + 1);
    HIP_CHECK(hipMemcpyAsync(v.data(), sums.p, sizeof(double) * dt->d, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    scratch.release(); sums.release();
    v[dt->d] = (double)dt->n;
    ctx->allreduce_host(v.data(), v.size());
    dt->n_global = (uint64_t)std::llround(v[dt->d]);
    dt->shift.resize(dt->d);
    for (int j = 0; j < dt->d; ++j) dt->shift[j] = v[j] / v[dt->d];
    dt->shift_dev.reserve(sizeof(double) * dt->D);              // zero-padded to the kernels' dimension D
    HIP_CHECK(hipMemsetAsync(dt->shift_dev.p, 0, sizeof(double) * dt->D, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(dt->shift_dev.p, dt->shift.data(), sizeof(double) * dt->d, hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();
}


mlhip_data* upload_common(mlhip_ctx* ctx, const double* x, bool on_device, uint32_t d, uint64_t n, int64_t ld)
{
    require(ctx != nullptr, "null context");
    require(x != nullptr || n == 0, "null data");
    require(d >= 1, "At least one dimension required");
    require(ld >= (int64_t)d, "ld must be >= d");
    require(n < 0xffffff00ull, "shard too large (n must fit 32 bits, like the reference's unsigned int)");
    const int D = padded_dim((int)d);
    if (D < 0) throw Unsupported("dimension d > 4096 is not supported");
    ctx->use();
    auto* dt = new mlhip_data;
    try {
        dt->ctx = ctx;
        dt->attach_pool(&ctx->pool);
        dt->d = (int)d;
        dt->D = D;
        dt->n = (uint32_t)n;
        dt->n_pad = padded_samples(n);
        if (dt->n_pad == 0) dt->n_pad = kSampleTile;
        dt->ldx = dt->n_pad;
        dt->xt.reserve(sizeof(double) * dt->ldx * D);
        HIP_CHECK(hipMemsetAsync(dt->xt.p, 0, sizeof(double) * dt->ldx * D, ctx->stream));
        if (on_device) {
            launch_transpose_to_dim_major(x, ld, dt->d, n, dt->xt.as<double>(), dt->ldx, 0, ctx->stream);
        } else {
            // Pageable host memory: the caller's block is packed into two pinned staging buffers by the CPU (this also
            // removes the ld > d padding) while the previous chunk's H2D copy + transpose run on the stream. A direct
            // hipMemcpy from pageable memory reaches only ~3 GB/s on this platform; pinned chunks go at PCIe rate.
            // Chunks are sized by BYTES (ctx->stage_bytes, 128 MB: 2^19 samples at d = 32), whatever the dimension: two pinned and
            // two device buffers of that size, not 8 d 2^19 bytes each (16 GB at d = 4096).
            const uint64_t chunk = std::max<uint64_t>(256, ctx->stage_bytes / (sizeof(double) * d));
            const uint64_t cap = n < chunk ? (n ? n : 1) : chunk;
            DevBuf* stage = ctx->up_stage;
            PinnedBuf* pin = ctx->up_pin;
            struct Events {
                hipEvent_t e[2] = {nullptr, nullptr};
                ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
            } done;
            for (int b = 0; b < 2; ++b) {
                stage[b].reserve(sizeof(double) * d * cap);
                pin[b].reserve(sizeof(double) * d * cap);
                HIP_CHECK(hipEventCreateWithFlags(&done.e[b], hipEventDisableTiming));
            }
            int b = 0;
            for (uint64_t i0 = 0; i0 < n; i0 += chunk, b ^= 1) {
                const uint64_t c = (n - i0 < chunk) ? n - i0 : chunk;
                if (i0 >= 2 * chunk) HIP_CHECK(hipEventSynchronize(done.e[b]));   // staging pair b is free again
                double* dst = pin[b].as<double>();
                const double* src = x + (int64_t)i0 * ld;
                if (ld == (int64_t)d) {
                    copy_bytes(dst, src, sizeof(double) * d * c);
                } else {
                    for (uint64_t i = 0; i < c; ++i) std::memcpy(dst + i * d, src + (int64_t)i * ld, sizeof(double) * d);
                }
                HIP_CHECK(hipMemcpyAsync(stage[b].p, dst, sizeof(double) * d * c, hipMemcpyHostToDevice, ctx->stream));
                launch_transpose_to_dim_major(stage[b].as<double>(), d, dt->d, c, dt->xt.as<double>(), dt->ldx, i0, ctx->stream);
                HIP_CHECK(hipEventRecord(done.e[b], ctx->stream));
            }
            ctx->sync();
        }
        finish_upload(dt);
    } catch (...) {
        delete dt;
        throw;
    }
    ctx->adopt(dt);
    return dt;
}


/// Device -> pageable host copy of `cols` columns of `col_bytes` bytes each (source / destination pitches given), staged through
/// the context's two pinned buffers: the CPU unpacks chunk i while the DMA engine fetches chunk i+1. A direct copy into
/// pageable memory runs at ~3 GB/s on this platform; this one at PCIe rate.
void download_columns(mlhip_ctx* ctx, char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t col_bytes, size_t cols)
{
    if (!col_bytes || !cols) return;
    if (col_bytes * cols <= (size_t(1) << 20)) {   // small: not worth the pipeline
        if (cols == 1)
            HIP_CHECK(hipMemcpyAsync(dst, src, col_bytes, hipMemcpyDeviceToHost, ctx->stream));
        else
            HIP_CHECK(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, col_bytes, cols, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        return;
    }
    const size_t chunk = size_t(64) << 20;
    PinnedBuf* pin = ctx->up_pin;
    hipEvent_t done[2];
    for (int b = 0; b < 2; ++b) {
        pin[b].reserve(std::min(chunk, col_bytes));
        HIP_CHECK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    }
    struct Piece { char* dst; size_t bytes; };
    Piece pending[2] = {{nullptr, 0}, {nullptr, 0}};
    int b = 0;
    for (size_t c = 0; c < cols; ++c)
        for (size_t off = 0; off < col_bytes; off += chunk, b ^= 1) {
            if (pending[b].bytes) {                     // unpack what this buffer held two pieces ago
                HIP_CHECK(hipEventSynchronize(done[b]));
                copy_bytes(pending[b].dst, pin[b].p, pending[b].bytes);
            }
            const size_t bytes = std::min(chunk, col_bytes - off);
            HIP_CHECK(hipMemcpyAsync(pin[b].p, src + c * src_pitch + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipEventRecord(done[b], ctx->stream));
            pending[b] = {dst + c * dst_pitch + off, bytes};
        }
    for (int k = 0; k < 2; ++k, b ^= 1)
        if (pending[b].bytes) {
            HIP_CHECK(hipEventSynchronize(done[b]));
            copy_bytes(pending[b].dst, pin[b].p, pending[b].bytes);
        }
    for (int k = 0; k < 2; ++k) (void)hipEventDestroy(done[k]);
}

}  // namespace mlhip_rt

extern "C" {


int mlhip_data_upload(mlhip_ctx* ctx, const double* x, uint32_t d, uint64_t n, int64_t ld, mlhip_data** out)
{
    return guarded([&] {
        require(out && ctx, "null argument");
        *out = ctx->group ? grp::upload(ctx, x, false, d, n, ld) : upload_common(ctx, x, false, d, n, ld);
    });
}
int mlhip_data_upload_dev(mlhip_ctx* ctx, const double* x_dev, uint32_t d, uint64_t n, int64_t ld, mlhip_data** out)
{
    return guarded([&] {
        require(out && ctx, "null argument");
        // (a group: x_dev must be readable from every shard's GPU -- the same GPU, or peers with access enabled)
        *out = ctx->group ? grp::upload(ctx, x_dev, true, d, n, ld) : upload_common(ctx, x_dev, true, d, n, ld);
    });
}
int mlhip_data_free(mlhip_data* data)
{
    return guarded([&] {
        if (!data) return;
        if (!data->ctx) { delete data; return; }                                      // (its context is gone: detached, nothing in flight)
        if (!data->parts.empty() || data->ctx->group) { data->ctx->disown(data); delete data; return; }   // (a group's block: its parts free themselves)
        (void)hipSetDevice(data->ctx->device);
        (void)hipStreamSynchronize(data->ctx->stream);
        data->ctx->disown(data);
        delete data;
    });
}
int mlhip_data_shape(const mlhip_data* data, uint32_t* d, uint64_t* n_local, uint64_t* n_global)
{
    return guarded([&] {
        require(data, "null data");
        if (d) *d = (uint32_t)data->d;
        if (n_local) *n_local = data->parts.empty() ? (uint64_t)data->n : data->n_global;   // (a group holds the whole sample)
        if (n_global) *n_global = data->n_global;
    });
}
int mlhip_data_shard_rows(const mlhip_data* data, int shard, uint64_t* first_row, uint64_t* n_rows)
{
    return guarded([&] {
        require(data, "null data");
        const int shards = data->parts.empty() ? 1 : (int)data->parts.size();
        require(shard >= 0 && shard < shards, "no such shard");
        if (first_row) *first_row = data->parts.empty() ? 0 : data->first_row[(size_t)shard];
        if (n_rows) *n_rows = data->parts.empty() ? (uint64_t)data->n : data->first_row[(size_t)shard + 1] - data->first_row[(size_t)shard];
    });
}
int mlhip_data_shift(const mlhip_data* data, double* shift)
{
    return guarded([&] {
        require(data && shift, "null argument");
        std::memcpy(shift, data->shift.data(), sizeof(double) * data->d);
    });
}

int mlhip_sample_covariance(mlhip_ctx* ctx, mlhip_data* data, double* mean, double* covariance)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::sample_covariance(ctx, data, mean, covariance); return; }
        check_em_args(ctx, data, 1);
        require(covariance, "null argument");
        const int saved_K = data->em_K;
        (void)saved_K;
        // K = 1, r = 1: S_0 = sum_i xt_i xt_i^T about the global mean. The E-step workspace for another K is
        // left untouched only if K == 1; otherwise it is rebuilt on the next E-step.
        ensure_em_workspace(data, 1);
        data->have_estep = false;
        launch_fill_responsibilities(nullptr, data->n, 1, data->lw.as<double>(), data->ldr, ctx->stream);
        run_mstats(data, 1, kFromResp, data->lw.as<double>(), data->ldr, false);
        const double* s = data->stats_host.as<double>();
        const int d = data->d;
        const double n = (double)data->n_global;
        // shift == global mean, so S1' is rounding noise; subtract its (tiny) contribution anyway.
        for (int a = 0; a < d; ++a) {
            const double ma = s[stats_index(d, a)] / n;
            if (mean) mean[a] = data->shift[a] + ma;
            for (int b = 0; b <= a; ++b) {
                const double v = (s[stats_index(a, b)] - s[stats_index(d, a)] * (s[stats_index(d, b)] / n)) / (n - 1.0);
                covariance[(size_t)b * d + a] = v;
                covariance[(size_t)a * d + b] = v;
            }
        }
    });
}

int mlhip_xxt_xy(mlhip_ctx* ctx, mlhip_data* data, const double* y, double* xxt, double* xy)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::xxt_xy(ctx, data, y, xxt, xy); return; }
        check_em_args(ctx, data, 2);
        require((y || data->n == 0) && xxt && xy, "null argument");
        ensure_em_workspace(data, 2);
        data->have_estep = false;
        // weight rows: [0] = 1 (valid samples), [1] = y; the statistics kernel then yields, about the shift s,
        //   component 0: N, sum (x - s), sum (x - s)(x - s)^T      component 1: sum y, sum y (x - s)
        double* w = data->lw.as<double>();
        launch_fill_responsibilities(nullptr, data->n, 1, w, data->ldr, ctx->stream);
        HIP_CHECK(hipMemsetAsync(w + data->ldr, 0, sizeof(double) * data->ldr, ctx->stream));
        if (data->n)
            HIP_CHECK(hipMemcpyAsync(w + data->ldr, y, sizeof(double) * data->n, hipMemcpyHostToDevice, ctx->stream));
        run_mstats(data, 2, kFromResp, w, data->ldr, false);
        const int d = data->d, F = stats_count(d);
        const double* s0 = data->stats_host.as<double>();
        const double* s1 = s0 + F;
        const double* sh = data->shift.data();
        const double n = s0[stats_index(d, d)], sum_y = s1[stats_index(d, d)];
        for (int a = 0; a < d; ++a) {
            xy[a] = s1[stats_index(d, a)] + sh[a] * sum_y;
            for (int b = 0; b <= a; ++b) {
                const double v = s0[stats_index(a, b)] + sh[a] * s0[stats_index(d, b)] + s0[stats_index(d, a)] * sh[b] +
                                 n * sh[a] * sh[b];
                xxt[(size_t)b * d + a] = v;
                xxt[(size_t)a * d + b] = v;
            }
        }
    });
}

int mlhip_random_partition_means(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* order, const uint32_t* offsets,
                                 double* means, double* sizes)
{
    return guarded([&] {
        if (ctx && ctx->group) { grp::random_partition_means(ctx, data, K, order, offsets, means, sizes); return; }
        check_em_args(ctx, data, K);
        require(offsets && means && sizes && (order || data->n == 0), "null argument");
        require(offsets[0] == 0 && offsets[K] == data->n, "offsets must cover this rank's rows");
        for (uint32_t k = 0; k < K; ++k) require(offsets[k] <= offsets[k + 1], "offsets must ascend");
        const int d = data->d;
        DevBuf order_dev, small;   // released below (one initialisation per fit: no point in keeping them)
        struct Release { DevBuf& a; DevBuf& b; ~Release() { a.release(); b.release(); } } release{order_dev, small};
        const size_t off_bytes = ((sizeof(uint32_t) * (K + 1) + 15) / 16) * 16;
        const size_t mean_doubles = (size_t)K * d;
        order_dev.reserve(std::max<size_t>(16, sizeof(uint32_t) * data->n));
        small.reserve(off_bytes + sizeof(double) * (mean_doubles + K));
        uint32_t* off_dev = small.as<uint32_t>();
        double* means_dev = reinterpret_cast<double*>(small.as<char>() + off_bytes);
        double* sizes_dev = means_dev + mean_doubles;
        if (data->n) HIP_CHECK(hipMemcpyAsync(order_dev.p, order, sizeof(uint32_t) * data->n, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(off_dev, offsets, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(means_dev, means, sizeof(double) * mean_doubles, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(sizes_dev, sizes, sizeof(double) * K, hipMemcpyHostToDevice, ctx->stream));
        ctx->timed("random_partition", [&] {
            launch_random_partition(data->xt.as<double>(), data->ldx, d, (int)K, order_dev.as<uint32_t>(), off_dev, means_dev, sizes_dev,
                                    ctx->stream);
        });
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(means, means_dev, sizeof(double) * mean_doubles, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(sizes, sizes_dev, sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
    });
}

}  // extern "C"
