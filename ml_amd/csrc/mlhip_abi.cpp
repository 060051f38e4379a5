// Runtime + C ABI (include/mlhip.h): device context, HBM-resident data, and the per-iteration
// orchestration of the gfx950 kernels. Host-side work here is only the tiny per-component d x d algebra
// (Cholesky / inverse / M-step closing arithmetic) that the reference also keeps out of its hot loops.
// There is no CPU fallback: every compute entry point needs a HIP device.
#include "mlhip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is dlopen'ed on first use (librccl is 570 MB; single-GPU users never pay for it)

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <ctime>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "device/device.hpp"
#include "host/em_math.hpp"

namespace {

thread_local std::string g_error;

struct InvalidArgument : std::runtime_error { using std::runtime_error::runtime_error; };
struct NoDevice : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
struct DomainError : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);  \
    } while (0)

template <class F> int guarded(F&& f)
{
    try { f(); return MLHIP_OK; }
    catch (const InvalidArgument& e) { g_error = e.what(); return MLHIP_E_INVALID_ARGUMENT; }
    catch (const DomainError& e) { g_error = e.what(); return MLHIP_E_DOMAIN; }
    catch (const NoDevice& e) { g_error = e.what(); return MLHIP_E_NO_DEVICE; }
    catch (const Unsupported& e) { g_error = e.what(); return MLHIP_E_UNSUPPORTED; }
    catch (const std::exception& e) { g_error = e.what(); return MLHIP_E_RUNTIME; }
}

/// Growable device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void reserve(size_t b)
    {
        if (b <= bytes) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr; bytes = 0;
        HIP_CHECK(hipMalloc(&p, b));
        bytes = b;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void reserve(size_t b)
    {
        if (b <= bytes) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr; bytes = 0;
        HIP_CHECK(hipHostMalloc(&p, b, hipHostMallocDefault));
        bytes = b;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct Timer {
    double total_ms = 0;
    uint64_t launches = 0;
};

/// RCCL entry points, resolved from librccl.so.1 the first time a communicator is asked for. A process that already
/// holds an RCCL (e.g. the copy bundled with PyTorch-ROCm, same soname) gets that one back from dlopen.
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;

    static Rccl& get()
    {
        static Rccl r = [] {
            Rccl x;
            const char* env = std::getenv("MLHIP_RCCL_LIBRARY");
            const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) {
                if (!n || !*n) continue;
                x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (x.handle) break;
            }
            if (!x.handle) return x;
            auto sym = [&](const char* name) { return dlsym(x.handle, name); };
            x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
            x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
            x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
            x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
            x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
            x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(sym("ncclGetVersion"));
            if (!(x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.CommCount && x.AllReduce && x.GetErrorString)) {
                dlclose(x.handle);
                x.handle = nullptr;
            }
            return x;
        }();
        if (!r.handle)
            throw std::runtime_error("RCCL is not available: librccl.so.1 could not be loaded (set MLHIP_RCCL_LIBRARY)");
        return r;
    }
    void check(ncclResult_t rc, const char* what) const
    {
        if (rc != ncclSuccess) throw std::runtime_error(std::string("RCCL error in ") + what + ": " + GetErrorString(rc));
    }
};

}  // namespace

struct mlhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    // all-reduce hook
    mlhip_allreduce_fn reduce_fn = nullptr;
    void* reduce_user = nullptr;
    int reduce_on_device = 0, world_size = 1, rank = 0;
    ncclComm_t comm = nullptr;   // library-owned RCCL communicator (mlhip_ctx_init_rccl); its all-reduce is the hook then
    // scratch
    DevBuf small_dev;        // for all-reducing short host vectors through a device hook
    PinnedBuf small_host;
    DevBuf up_stage[2];      // upload staging (kept across uploads: pinned allocations are slow)
    PinnedBuf up_pin[2];
    // timing
    bool timing = false;
    typedef std::pair<hipEvent_t, hipEvent_t> EventPair;
    std::vector<EventPair> spare_events;
    std::vector<std::pair<const char*, EventPair>> pending;     // (names are string literals)
    std::map<std::string, Timer> timers;

    void use() const { HIP_CHECK(hipSetDevice(device)); }
    void sync() const { HIP_CHECK(hipStreamSynchronize(stream)); }

    /// Device time of a launch by a pair of HIP events on the stream the kernel goes to. The pair is only RECORDED here;
    /// the elapsed times are read when somebody asks (resolve_timers), so a timed region runs as it does untimed: no
    /// synchronisation between launches, the clocks the chip holds under a back-to-back stream of kernels.
    template <class F> void timed(const char* name, F&& launch)
    {
        if (!timing) { launch(); return; }
        if (pending.size() >= 4096) resolve_timers();
        EventPair e;
        if (!spare_events.empty()) { e = spare_events.back(); spare_events.pop_back(); }
        else { HIP_CHECK(hipEventCreate(&e.first)); HIP_CHECK(hipEventCreate(&e.second)); }
        HIP_CHECK(hipEventRecord(e.first, stream));
        launch();
        HIP_CHECK(hipEventRecord(e.second, stream));
        pending.push_back({name, e});
    }

    void resolve_timers()
    {
        if (pending.empty()) return;
        HIP_CHECK(hipEventSynchronize(pending.back().second.second));
        for (auto& p : pending) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, p.second.first, p.second.second));
            Timer& t = timers[p.first];
            t.total_ms += ms;
            t.launches += 1;
            spare_events.push_back(p.second);
        }
        pending.clear();
    }

    /// The installed hook on a DEVICE buffer, on the context's stream. Timed as "allreduce" (event pair around the collective
    /// on the stream: on a rank that arrives early this includes the wait for the slowest rank -- what a first multi-GPU run
    /// needs to see).
    void reduce_device(double* buf, size_t count)
    {
        int rc = 0;
        timed("allreduce", [&] { rc = reduce_fn(reduce_user, buf, count, 1, stream); });
        if (rc != 0) throw std::runtime_error("all-reduce hook failed");
    }

    /// End-of-fit guard of a row-sharded job: parameters are never broadcast -- every rank applies the same closing arithmetic
    /// to the same all-reduced sums -- so ranks that received different sums (a collective that is not bitwise reproducible
    /// across ranks, a rank on different data) would drift apart silently. Every rank puts a 48-bit checksum of its results
    /// (three exactly representable 16-bit pieces) into its own slot of a zero vector, the vector is summed across ranks,
    /// and every rank compares all slots. One small collective per fit. MLHIP_RANK_CHECK=0 disables.
    void check_ranks_agree(const char* what, std::initializer_list<std::pair<const double*, size_t>> blocks)
    {
        if (!reduce_fn || world_size <= 1) return;
        static const bool on = [] { const char* e = std::getenv("MLHIP_RANK_CHECK"); return !(e && e[0] == '0'); }();
        if (!on) return;
        uint64_t h = 1469598103934665603ull;                         // FNV-1a over the bytes of the blocks
        for (const auto& b : blocks) {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(b.first);
            for (size_t i = 0; i < b.second * sizeof(double); ++i) { h ^= p[i]; h *= 1099511628211ull; }
        }
        std::vector<double> v(3 * (size_t)world_size, 0.0);
        for (int j = 0; j < 3; ++j) v[3 * (size_t)rank + j] = (double)((h >> (16 * j)) & 0xffffu);
        allreduce_host(v.data(), v.size());
        for (int r = 1; r < world_size; ++r)
            for (int j = 0; j < 3; ++j)
                if (v[3 * (size_t)r + j] != v[j])
                    throw std::runtime_error(std::string("ranks disagree on ") + what + " at the end of the fit (rank " + std::to_string(r) +
                                             " differs from rank 0): the statistics all-reduce did not give every rank the same sums");
    }

    /// Sum `count` host doubles across ranks (no-op single rank).
    void allreduce_host(double* v, size_t count)
    {
        if (!reduce_fn) return;
        if (reduce_on_device) {
            small_dev.reserve(count * sizeof(double));
            HIP_CHECK(hipMemcpyAsync(small_dev.p, v, count * sizeof(double), hipMemcpyHostToDevice, stream));
            reduce_device(small_dev.as<double>(), count);
            HIP_CHECK(hipMemcpyAsync(v, small_dev.p, count * sizeof(double), hipMemcpyDeviceToHost, stream));
            sync();
        } else {
            if (reduce_fn(reduce_user, v, count, 0, stream) != 0) throw std::runtime_error("all-reduce hook failed");
        }
    }
};

struct mlhip_data {
    mlhip_ctx* ctx = nullptr;
    int d = 0, D = 0;
    uint32_t n = 0, n_pad = 0;
    uint64_t n_global = 0;
    size_t ldx = 0;
    DevBuf xt;                    // [D][ldx]
    DevBuf shift_dev;             // d doubles
    std::vector<double> shift;    // host copy
    // EM workspace (sized for em_K)
    int em_K = 0;
    size_t ldr = 0;
    DevBuf lw, lse, esum, ll_partials, params_dev, partials, stats_dev, resp_dev, labels_dev;
    PinnedBuf params_host, stats_host;
    int n_ll = 0;
    bool have_estep = false;
    bool lw_valid = false;        // false after a fused step: lw is rebuilt from params_dev on demand (ensure_lw)
    int estep_variant = 0;        // record layout currently in params_dev: 0 = valu, 1 = mfma16, 2 = mfma4
    bool estep_fold = false;      // mfma4 records in FOLD form (vector slot = -W (mu - shift)): layout.hpp kEstepFoldLimit
    // diagonal-covariance extension: parameters of the last mlhip_em_step_diag (the N x K block is rebuilt from them on demand)
    bool diag_step = false;
    std::vector<double> diag_mixing, diag_means, diag_vars;
    // mlhip_em_iterate: parameters and the next E-step's records stay on the device between iterations
    DevBuf params_next, it_pack[3];      // it_pack: [info (1 + 2K) | mixing (K) | means (K d) | covariances], one D2H covers it
    PinnedBuf it_info_host;
    // ... and, for the lagged (speculative) loop of small shapes: a third record buffer, one read-back slot and event per pack
    DevBuf params_prev;
    PinnedBuf it_info_slot[3];
    hipEvent_t it_event[3] = {nullptr, nullptr, nullptr};
    // source of the last statistics pass (for the per-component refinement pass)
    int stats_mode = 0;
    const double* stats_resp = nullptr;
    size_t stats_ld = 0;
    DevBuf refine_shift, refine_stats;
    uint64_t refined_components = 0;   // diagnostic counter
    // K-means workspace
    DevBuf km_labels[2], km_cent, km_cent_next, km_partials, km_out, km_mind, km_probe, km_scale, km_cnorm, km_xt_pad;
    PinnedBuf km_host;
    int km_cur = 0;
    bool km_have_old = false;

    ~mlhip_data()
    {
        for (DevBuf* b : {&xt, &shift_dev, &lw, &lse, &esum, &ll_partials, &params_dev, &partials, &stats_dev, &resp_dev,
                          &labels_dev, &km_labels[0], &km_labels[1], &km_cent, &km_cent_next, &km_partials, &km_out, &km_mind, &km_probe, &km_scale, &km_cnorm, &km_xt_pad,
                          &refine_shift, &refine_stats, &params_next, &params_prev, &it_pack[0], &it_pack[1], &it_pack[2]})
            b->release();
        it_info_host.release();
        for (auto& sl : it_info_slot) sl.release();
        for (auto& e : it_event) if (e) (void)hipEventDestroy(e);
        params_host.release(); stats_host.release(); km_host.release();
    }
};

namespace {

using namespace mlhip;

constexpr int kMaxLlPartials = 2048;

/// MLHIP_TRACE=1: wall-clock microseconds of the host-visible phases of one EM iteration on stderr.
struct PhaseTrace {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseTrace() : on([] { const char* e = std::getenv("MLHIP_TRACE"); return e && e[0] == '1'; }()), t(std::chrono::steady_clock::now()) {}
    void mark(const char* name)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mlhip] %-22s %8.1f us\n", name, std::chrono::duration<double, std::micro>(now - t).count());
        t = now;
    }
};

void require(bool ok, const char* msg) { if (!ok) throw InvalidArgument(msg); }

int env_int(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}

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
    if (D < 0) throw Unsupported("dimension d > 128 is not supported by the register-resident kernels yet");
    ctx->use();
    auto* dt = new mlhip_data;
    try {
        dt->ctx = ctx;
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
            const uint64_t chunk = 1u << 19;                       // samples per chunk (128 MB at d = 32)
            const uint64_t cap = n < chunk ? (n ? n : 1) : chunk;
            DevBuf* stage = ctx->up_stage;
            PinnedBuf* pin = ctx->up_pin;
            hipEvent_t done[2];
            for (int b = 0; b < 2; ++b) {
                stage[b].reserve(sizeof(double) * d * cap);
                pin[b].reserve(sizeof(double) * d * cap);
                HIP_CHECK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
            }
            int b = 0;
            for (uint64_t i0 = 0; i0 < n; i0 += chunk, b ^= 1) {
                const uint64_t c = (n - i0 < chunk) ? n - i0 : chunk;
                if (i0 >= 2 * chunk) HIP_CHECK(hipEventSynchronize(done[b]));   // staging pair b is free again
                double* dst = pin[b].as<double>();
                const double* src = x + (int64_t)i0 * ld;
                if (ld == (int64_t)d) {
                    copy_bytes(dst, src, sizeof(double) * d * c);
                } else {
                    for (uint64_t i = 0; i < c; ++i) std::memcpy(dst + i * d, src + (int64_t)i * ld, sizeof(double) * d);
                }
                HIP_CHECK(hipMemcpyAsync(stage[b].p, dst, sizeof(double) * d * c, hipMemcpyHostToDevice, ctx->stream));
                launch_transpose_to_dim_major(stage[b].as<double>(), d, dt->d, c, dt->xt.as<double>(), dt->ldx, i0, ctx->stream);
                HIP_CHECK(hipEventRecord(done[b], ctx->stream));
            }
            ctx->sync();
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(done[k]);
        }
        finish_upload(dt);
    } catch (...) {
        delete dt;
        throw;
    }
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
void prepare_estep(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs, DevBuf* target = nullptr)
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
    if (use_mfma4) {
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
void launch_estep(mlhip_data* dt, int K, bool with_lse = true)
{
    mlhip_ctx* ctx = dt->ctx;
    EstepArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.D = dt->D;
    a.params = dt->params_dev.as<double>(); a.K = K;
    a.lw = dt->lw.as<double>(); a.ldr = dt->ldr; a.lse = dt->lse.as<double>();
    a.ll_partials = dt->ll_partials.as<double>(); a.n_ll_partials = kMaxLlPartials;
    a.shift = dt->shift_dev.as<double>(); a.fold = dt->estep_fold ? 1 : 0;
    a.with_lse = (with_lse || dt->estep_variant != 2) ? 1 : 0;
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

void run_estep(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs, bool with_lse = true)
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
void collect_stats(mlhip_data* dt, int K, size_t count = 0)
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
            throw std::runtime_error("all-reduce hook failed");
    }
}

/// One EM iteration's device work in a single kernel where the shape allows (d <= 6, K <= 32 or d <= 4, K <= 64: em_fused_small.hip): no
/// N x K block in HBM. MLHIP_FUSED=0 keeps the two-kernel path. Returns false when the shape is not covered.
bool fused_step_applies(const mlhip_data* dt, int K)
{
    const char* env = std::getenv("MLHIP_FUSED");
    return !(env && env[0] == '0') && mstats::em_fused_supported(dt->d, K);
}

/// The fused kernel + reduction on the records already in params_dev; statistics end in stats_dev (and, with `collect`, all-
/// reduced in stats_host).
void launch_fused_step(mlhip_data* dt, int K, bool collect)
{
    mlhip_ctx* ctx = dt->ctx;
    FusedArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.d = dt->d;
    a.shift = dt->shift_dev.as<double>(); a.params = dt->params_dev.as<double>(); a.K = K;
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
void run_mstats(mlhip_data* dt, int K, int mode, const double* resp_dev, size_t ld_resp, bool with_ll, bool collect = true)
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
            throw std::runtime_error("all-reduce hook failed");
    }
    HIP_CHECK(hipMemcpyAsync(s.data(), dt->refine_stats.p, sizeof(double) * F, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->reduce_fn && !ctx->reduce_on_device) {
        if (ctx->reduce_fn(ctx->reduce_user, s.data(), (size_t)F, 0, ctx->stream) != 0) throw std::runtime_error("all-reduce hook failed");
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
void run_diag_kernel(mlhip_data* dt, int K, const double* shift_dev, bool collect = true)
{
    mlhip_ctx* ctx = dt->ctx;
    DiagArgs a{};
    a.xt = dt->xt.as<double>(); a.ldx = dt->ldx; a.n = dt->n; a.d = dt->d;
    a.shift = shift_dev; a.params = dt->params_dev.as<double>(); a.K = K;
    a.lse = dt->lse.as<double>();
    a.partials = dt->partials.as<double>(); a.partials_capacity = dt->partials.bytes / sizeof(double);
    a.ll_partials = dt->ll_partials.as<double>(); a.n_ll_partials = kMaxLlPartials;
    int grid = 0;
    ctx->timed("em_diag", [&] { grid = mstats::launch_em_diag(a, ctx->num_cus, ctx->stream); });
    if (grid <= 0) throw std::runtime_error("diagonal EM kernel launch failed");
    launch_em_reduce_blocks(a.partials, grid, mstats::em_diag_partial_rows(K), mstats::em_diag_partial_cols(dt->d), K,
                            diag_stats_count(dt->d), a.ll_partials, grid, dt->stats_dev.as<double>(), ctx->stream);
    HIP_CHECK(hipGetLastError());
    dt->n_ll = grid;
    if (collect) collect_stats(dt, K, (size_t)K * diag_stats_count(dt->d) + 1);
}

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

/// What one K-means pass runs on: the data block (a zero-padded copy where the matrix-core kernel needs one) and its rows.
struct KmBlock {
    const double* xt;
    int D;
};

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
void km_launch(mlhip_data* dt, int K, const KmBlock& b, bool accumulate, double* min_dist_out)
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
    launch_kmeans_reduce(a, rc, ctx->stream);
    HIP_CHECK(hipGetLastError());
    dt->km_cur = nxt;
    dt->km_have_old = true;
    if (ctx->reduce_fn && ctx->reduce_on_device) {
        const size_t count = 2 + (accumulate ? (size_t)K * (dt->d + 1) : 0);
        ctx->reduce_device(dt->km_out.as<double>(), count);
    }
}

/// km_out -> km_host (`count` doubles), summed across ranks on the host when the all-reduce works on host memory.
void km_fetch(mlhip_data* dt, size_t count)
{
    mlhip_ctx* ctx = dt->ctx;
    double* ch = dt->km_host.as<double>();
    HIP_CHECK(hipMemcpyAsync(ch, dt->km_out.p, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->reduce_fn && !ctx->reduce_on_device) {
        if (ctx->reduce_fn(ctx->reduce_user, ch, count, 0, ctx->stream) != 0) throw std::runtime_error("all-reduce hook failed");
    }
}

/// Assignment (+ optional accumulation); leaves all-reduced [inertia, changed, counts, sums] in km_host.
void run_kmeans(mlhip_data* dt, int K, const double* centroids, bool accumulate, double* min_dist_out = nullptr)
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
    const bool device_route = !(ctx->reduce_fn && !ctx->reduce_on_device) && !std::getenv("MLHIP_KMEANS_HOST_LOOP");
    const KmBlock b = km_block(dt, K);
    std::vector<double> cur(centroids, centroids + kd), old(kd, 0.0), upd(kd);
    if (device_route) {
        dt->km_cent_next.reserve(sizeof(double) * (size_t)K * b.D);
        km_upload_centroids(dt, K, b, cur.data());
    }
    *converged = 0;
    *steps_done = 0;
    for (uint32_t step = 0; step < max_steps; ++step) {
        if (device_route) {
            km_launch(dt, K, b, true, nullptr);
            launch_kmeans_close(dt->km_out.as<double>(), K, d, b.D, dt->km_cent_next.as<double>(), ctx->stream);
            km_fetch(dt, 2 + (size_t)K * (d + 1));
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
        throw std::runtime_error("all-reduce hook failed");
    HIP_CHECK(hipMemcpyAsync(dt->stats_dev.p, dt->stats_host.p, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
}

/// Records of a diagonal-covariance parameter set -> `target` (padded to whole 16-component row blocks with neutral records).
void upload_diag_records(mlhip_data* data, int K, const double* mixing, const double* means, const double* variances, DevBuf& target)
{
    mlhip_ctx* ctx = data->ctx;
    const int KP = mstats::em_diag_partial_rows(K);
    const size_t rec_bytes = sizeof(double) * diag_param_stride(data->D) * (size_t)KP;
    target.reserve(rec_bytes);
    data->params_host.reserve(rec_bytes);
    host::build_diag_params(data->d, data->D, K, KP, mixing, means, variances, data->params_host.as<double>());
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

/// One diagonal-covariance EM iteration with the closing arithmetic on the HOST (the body of mlhip_em_step_diag).
void em_step_full(mlhip_data* data, int K, const double* mixing, const double* means, const double* covariances,
                  double* log_likelihood, double* mixing_out, double* means_out, double* covariances_out);

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

/// The loop of EM::fit (ML/EM.cpp:143-170) with everything between two convergence tests on the device: E-step, statistics,
/// all-reduce, closing arithmetic + next records (em_close.hip); per iteration the host reads back 1 + 2K doubles (log-
/// likelihood sum, refinement flags, FOLD criterion) and decides. A flagged component (far, tight cluster) sends that one
/// iteration through the host closing with its refinement pass, exactly as mlhip_em_step would. MLHIP_DEVICE_CLOSE=0, or
/// d > 64, runs the whole loop through the per-step functions.
void em_iterate(mlhip_data* data, int K, bool diag, double* mixing, double* means, double* covs, uint32_t max_steps, double atol,
                double rtol, uint32_t* steps_done, int* converged, double* log_likelihood, double* history)
{
    mlhip_ctx* ctx = data->ctx;
    const int d = data->d;
    *steps_done = 0;
    *converged = 0;
    double old_ll = -HUGE_VAL;
    auto test = [&](uint32_t step, double ll) {       // ML/EM.cpp:161-168
        if (history) history[step] = ll;
        *log_likelihood = ll;
        *steps_done = step + 1;
        if (step > 0) {
            const double change = std::fabs(ll - old_ll);
            if (change < atol + rtol * std::max(std::fabs(old_ll), std::fabs(ll))) { *converged = 1; return true; }
        }
        old_ll = ll;
        return false;
    };
    static const bool device_close_allowed = [] { const char* e = std::getenv("MLHIP_DEVICE_CLOSE"); return !(e && e[0] == '0'); }();
    ensure_em_workspace(data, K);
    bool device_close = device_close_allowed && em_close_supported(d) && !(diag && !mstats::em_diag_supported(d, K));
    if (device_close && !diag) {
        prepare_estep(data, K, mixing, means, covs);           // records of the caller's parameters -> params_dev
        if (data->estep_variant == 1) device_close = false;    // (experimental record layout: host closing only)
    }
    if (!device_close) {
        for (uint32_t step = 0; step < max_steps; ++step) {
            double ll = 0;
            if (diag) em_step_diag(data, K, mixing, means, covs, &ll, mixing, means, covs);
            else em_step_full(data, K, mixing, means, covs, &ll, mixing, means, covs);
            if (test(step, ll)) break;
        }
        return;
    }

    const size_t n_cov = diag ? (size_t)K * d : (size_t)K * d * d;
    const size_t F = diag ? diag_stats_count(d) : stats_count(d);
    const size_t n_info = em_close_info_doubles(K);
    const size_t n_pack = n_info + K + (size_t)K * d + n_cov;
    for (int b = 0; b < 3; ++b) data->it_pack[b].reserve(sizeof(double) * n_pack);
    data->it_info_host.reserve(sizeof(double) * n_pack);          // info, then (diagonal mode) a shadow of the newest parameters
    auto pack_mixing = [&](int b) { return data->it_pack[b].as<double>() + n_info; };
    auto pack_means = [&](int b) { return pack_mixing(b) + K; };
    auto pack_covs = [&](int b) { return pack_means(b) + (size_t)K * d; };
    if (diag) {
        upload_diag_records(data, K, mixing, means, covs, data->params_dev);
        upload_diag_records(data, K, mixing, means, covs, data->params_next);          // (the neutral padding records live in both)
        data->diag_mixing.assign(mixing, mixing + K);
        data->diag_means.assign(means, means + (size_t)K * d);
        data->diag_vars.assign(covs, covs + (size_t)K * d);
    } else {
        data->params_next.reserve(data->params_dev.bytes);
    }
    data->diag_step = diag;
    const bool fused = !diag && data->estep_variant == 0 && fused_step_applies(data, K);
    const bool self_norm = !diag && !fused && data->estep_variant == 2 && self_norm_applies(data, K);
    static const bool fold_allowed = [] { const char* e = std::getenv("MLHIP_ESTEP_FOLD"); return !(e && e[0] == '0'); }();
    const double limit = refine_ratio();
    std::vector<double> prev_mixing, prev_means, prev_vars;     // diag: the inputs of the E-step before the newest parameters
    int cur = 0;
    bool latest_on_host = true;
    double* info = data->it_info_host.as<double>();
    double* shadow = info + n_info;

    // One iteration's device work: E-step + statistics from the records in params_dev, all-reduce, closing arithmetic into
    // it_pack[out] and the next records into params_next. Nothing here waits for the device.
    auto launch_iteration = [&](int out) {
        if (diag) {
            run_diag_kernel(data, K, data->shift_dev.as<double>(), false);
        } else if (fused) {
            launch_fused_step(data, K, false);
        } else {
            launch_estep(data, K, !self_norm);
            run_mstats(data, K, self_norm ? kFromLogRespSelfNorm : kFromLogResp, nullptr, 0, true, false);
        }
        data->have_estep = true;
        data->lw_valid = !(diag || fused);
        allreduce_stats_dev(data, (size_t)K * F + 1);
        CloseArgs ca{};
        ca.stats = data->stats_dev.as<double>(); ca.K = K; ca.d = d; ca.D = data->D;
        ca.shift = data->shift_dev.as<double>(); ca.n_global = (double)data->n_global;
        ca.layout = data->estep_variant; ca.refine_limit = limit;
        ca.mixing = pack_mixing(out); ca.means = pack_means(out);
        ca.covs = pack_covs(out); ca.records = data->params_next.as<double>();
        ca.info = data->it_pack[out].as<double>();
        ctx->timed("em_close", [&] { if (diag) launch_em_close_diag(ca, ctx->stream); else launch_em_close(ca, ctx->stream); });
        HIP_CHECK(hipGetLastError());
    };

    // ---- lagged loop (small shapes: an iteration is tens of microseconds, of which the host's launches and its wait for the
    // read-back are most). Iteration i + 1 is launched BEFORE the host looks at iteration i's log-likelihood: the convergence
    // test of ML/EM.cpp:161-168 then fires one iteration late, and the speculative iteration is simply dropped -- three record
    // buffers and three packs keep the inputs and outputs of iteration i intact while i + 1 runs, so the results are
    // bit-identical to the synchronous loop. Not taken when the host has to decide something per iteration (FOLD form of the
    // matrix-core E-step) or carries the all-reduce itself (host hooks); a refinement flag (far, tight component) rolls the
    // loop back to the flagged iteration and hands over to the synchronous loop below. MLHIP_LAGGED=0: off.
    static const bool lagged_allowed = [] { const char* e = std::getenv("MLHIP_LAGGED"); return !(e && e[0] == '0'); }();
    const bool lagged = lagged_allowed && data->estep_variant != 2 && (!ctx->reduce_fn || ctx->reduce_on_device) && max_steps >= 2;
    uint32_t first_sync_step = 0;
    if (lagged) {
        const size_t copy_doubles = diag ? n_pack : n_info;
        for (int b = 0; b < 3; ++b) {
            data->it_info_slot[b].reserve(sizeof(double) * n_pack);
            if (!data->it_event[b]) HIP_CHECK(hipEventCreateWithFlags(&data->it_event[b], hipEventDisableTiming));
        }
        data->params_prev.reserve(data->params_dev.bytes);
        data->params_next.reserve(data->params_dev.bytes);
        std::vector<double> shadow_of[3];                        // diag: host copy of pack b's parameters (inputs of an E-step)
        if (diag) {
            // the neutral padding records must live in all three record buffers (params_next may just have been re-allocated)
            upload_diag_records(data, K, mixing, means, covs, data->params_next);
            upload_diag_records(data, K, mixing, means, covs, data->params_prev);
            shadow_of[0].assign(mixing, mixing + K);
            shadow_of[0].insert(shadow_of[0].end(), means, means + (size_t)K * d);
            shadow_of[0].insert(shadow_of[0].end(), covs, covs + n_cov);
        }
        auto launch = [&](uint32_t i) {                          // iteration i: records R_i (params_dev) -> R_(i+1), pack (i+1) % 3
            const int out = (int)((i + 1) % 3);
            launch_iteration(out);
            HIP_CHECK(hipMemcpyAsync(data->it_info_slot[out].p, data->it_pack[out].p, sizeof(double) * copy_doubles, hipMemcpyDeviceToHost,
                                     ctx->stream));
            HIP_CHECK(hipEventRecord(data->it_event[out], ctx->stream));
            // rotate: params_dev <- R_(i+1), params_prev <- R_i, params_next <- the buffer of R_(i-1) (evaluated, free)
            std::swap(data->params_prev, data->params_dev);      // prev = R_i, dev = old prev
            std::swap(data->params_dev, data->params_next);      // dev = R_(i+1), next = old prev
        };
        launch(0);
        uint32_t launched = 1;
        bool handed_over = false, stopped = false;
        uint32_t last = 0;
        for (uint32_t i = 0; i < max_steps; ++i) {
            if (i + 1 < max_steps) { launch(i + 1); launched = i + 2; }
            const int slot = (int)((i + 1) % 3);
            HIP_CHECK(hipEventSynchronize(data->it_event[slot]));
            const double* inf = data->it_info_slot[slot].as<double>();
            const double ll = inf[0] / (double)data->n_global - (double)d * log_two_pi() / 2;   // ML/EM.cpp:197-198, 211
            bool flagged = false;
            for (int k = 0; k < K; ++k) flagged = flagged || inf[1 + k] != 0.0;
            if (diag) shadow_of[slot].assign(inf + n_info, inf + n_info + K + (size_t)K * d + n_cov);
            last = i;
            if (flagged) {
                // roll back to the start of iteration i: records R_i into params_dev, parameters P_i into the caller's arrays
                ctx->sync();
                if (launched == i + 2) std::swap(data->params_dev, data->params_next);     // (next holds R_i after two rotations)
                else std::swap(data->params_dev, data->params_prev);
                if (i > 0) {
                    const int in = (int)(i % 3);
                    HIP_CHECK(hipMemcpyAsync(mixing, pack_mixing(in), sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipMemcpyAsync(means, pack_means(in), sizeof(double) * K * d, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipMemcpyAsync(covs, pack_covs(in), sizeof(double) * n_cov, hipMemcpyDeviceToHost, ctx->stream));
                    ctx->sync();
                }
                if (diag) {
                    const std::vector<double>& sh = shadow_of[i % 3];
                    data->diag_mixing.assign(sh.begin(), sh.begin() + K);
                    data->diag_means.assign(sh.begin() + K, sh.begin() + K + (size_t)K * d);
                    data->diag_vars.assign(sh.begin() + K + (size_t)K * d, sh.end());
                    upload_diag_records(data, K, mixing, means, covs, data->params_next);   // (its neutral padding records)
                }
                first_sync_step = i;
                handed_over = true;
                break;
            }
            if (test(i, ll) || i + 1 == max_steps) { stopped = true; break; }
        }
        if (!handed_over) {
            (void)stopped;
            ctx->sync();                                         // a speculative iteration may still be running: let it finish
            // device state as the synchronous loop leaves it: the records of the LAST evaluated E-step in params_dev; what the
            // speculative iteration overwrote (log-responsibilities, lse) is rebuilt from them on demand
            const bool speculated = launched == last + 2;
            if (speculated) { std::swap(data->params_dev, data->params_next); data->lw_valid = false; }
            else std::swap(data->params_dev, data->params_prev);
            const int res = (int)((last + 1) % 3);               // P_(last+1): the newest parameters
            HIP_CHECK(hipMemcpyAsync(mixing, pack_mixing(res), sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipMemcpyAsync(means, pack_means(res), sizeof(double) * K * d, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipMemcpyAsync(covs, pack_covs(res), sizeof(double) * n_cov, hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
            if (diag) {                                          // ensure_lw rebuilds the block from the inputs of the last E-step
                const std::vector<double>& sh = shadow_of[last % 3];
                data->diag_mixing.assign(sh.begin(), sh.begin() + K);
                data->diag_means.assign(sh.begin() + K, sh.begin() + K + (size_t)K * d);
                data->diag_vars.assign(sh.begin() + K + (size_t)K * d, sh.end());
            }
            return;
        }
    }

    for (uint32_t step = first_sync_step; step < max_steps; ++step) {
        PhaseTrace tr;
        const int nxt = cur ^ 1;
        launch_iteration(nxt);
        // one read-back: the info block and, in diagonal mode (small), a host shadow of the newest parameters right behind it
        // (ensure_lw needs the inputs of the last E-step)
        HIP_CHECK(hipMemcpyAsync(info, data->it_pack[nxt].p, sizeof(double) * (diag ? n_pack : n_info), hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        tr.mark("iteration (device close)");
        const double ll = info[0] / (double)data->n_global - (double)d * log_two_pi() / 2;   // ML/EM.cpp:197-198, 211
        bool flagged = false;
        double cmax = 0.0;
        for (int k = 0; k < K; ++k) {
            flagged = flagged || info[1 + k] != 0.0;
            cmax = std::max(cmax, info[1 + K + k]);
        }
        bool fold_next = false;
        if (flagged) {
            // a far, tight component: this iteration is closed on the host, refinement pass included (the per-step arithmetic)
            HIP_CHECK(hipMemcpyAsync(data->stats_host.p, data->stats_dev.p, sizeof(double) * ((size_t)K * F + 1),
                                     hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
            if (diag) {
                host::finalize_mstep_diag(d, K, data->stats_host.as<double>(), data->shift.data(), (double)data->n_global, mixing, means, covs);
                refine_diag(data, K, mixing, means, covs);
                upload_diag_records(data, K, mixing, means, covs, data->params_next);
                prev_mixing = data->diag_mixing; prev_means = data->diag_means; prev_vars = data->diag_vars;
                data->diag_mixing.assign(mixing, mixing + K);
                data->diag_means.assign(means, means + (size_t)K * d);
                data->diag_vars.assign(covs, covs + (size_t)K * d);
            } else {
                finalize_out(data, K, mixing, means, covs);
                const int variant = data->estep_variant;
                const bool fold_now = data->estep_fold;
                prepare_estep(data, K, mixing, means, covs, &data->params_next);
                fold_next = data->estep_fold;
                data->estep_fold = fold_now;                     // (still describes the records in params_dev)
                if (data->estep_variant != variant) throw std::runtime_error("E-step record layout changed inside a fit");
            }
            latest_on_host = true;
        } else {
            latest_on_host = false;
            cur = nxt;
            fold_next = fold_allowed && data->estep_variant == 2 && data->D <= kRegDim && cmax <= kEstepFoldLimit;
            if (diag) {
                prev_mixing = data->diag_mixing; prev_means = data->diag_means; prev_vars = data->diag_vars;
                data->diag_mixing.assign(shadow, shadow + K);
                data->diag_means.assign(shadow + K, shadow + K + (size_t)K * d);
                data->diag_vars.assign(shadow + K + (size_t)K * d, shadow + K + (size_t)K * d + n_cov);
            }
        }
        const bool stop = test(step, ll);
        if (stop || step + 1 == max_steps) break;
        std::swap(data->params_dev, data->params_next);          // the new records become the next E-step's
        data->estep_fold = fold_next;
    }
    // the caller's arrays receive the newest parameters; the device keeps the records of the LAST E-step in params_dev
    if (!latest_on_host) {
        HIP_CHECK(hipMemcpyAsync(mixing, pack_mixing(cur), sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(means, pack_means(cur), sizeof(double) * K * d, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(covs, pack_covs(cur), sizeof(double) * n_cov, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
    }
    if (diag && !prev_mixing.empty()) {   // ensure_lw rebuilds the block from the inputs of the last E-step
        data->diag_mixing = prev_mixing; data->diag_means = prev_means; data->diag_vars = prev_vars;
    }
}

/// The all-reduce hook of a context that owns an RCCL communicator: one ncclAllReduce(double, sum), in place, on the
/// context's stream -- ordered with the kernels before it and the copies after it, no host synchronisation.
int rccl_allreduce_hook(void* user, double* buf, size_t count, int on_device, void* stream)
{
    auto* ctx = static_cast<mlhip_ctx*>(user);
    if (!ctx || !ctx->comm || !on_device) return 1;
    const Rccl& r = Rccl::get();
    return r.AllReduce(buf, buf, count, ncclDouble, ncclSum, ctx->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

void drop_rccl(mlhip_ctx* ctx)
{
    if (!ctx->comm) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)Rccl::get().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    if (ctx->reduce_fn == rccl_allreduce_hook) {
        ctx->reduce_fn = nullptr; ctx->reduce_user = nullptr; ctx->reduce_on_device = 0; ctx->world_size = 1; ctx->rank = 0;
    }
}

void init_rccl(mlhip_ctx* ctx, const ncclUniqueId& id, int world_size, int rank)
{
    require(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size / rank");
    ctx->use();
    const Rccl& r = Rccl::get();
    drop_rccl(ctx);
    // RCCL prints its version banner on the C-level stdout when NCCL_DEBUG=VERSION/INFO is set; nothing else is written.
    r.check(r.CommInitRank(&ctx->comm, world_size, id, rank), "ncclCommInitRank");
    int count = 0;
    r.check(r.CommCount(ctx->comm, &count), "ncclCommCount");
    if (count != world_size) throw std::runtime_error("RCCL communicator size does not match world_size");
    ctx->reduce_fn = rccl_allreduce_hook;
    ctx->reduce_user = ctx;
    ctx->reduce_on_device = 1;
    ctx->world_size = world_size;
    ctx->rank = rank;
    int local = world_size;
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= world_size) local = v;
    }
    host::set_host_ranks(local);
}

}  // namespace

extern "C" {

const char* mlhip_last_error(void) { return g_error.c_str(); }
/* Internal: lets the C++ facade's C wrappers (mlpp_capi.cpp) report through the same slot. */
void mlhip_set_last_error_(const char* msg) { g_error = msg ? msg : ""; }
const char* mlhip_version(void) { return "0.1.0 (gfx950)"; }

int mlhip_device_count(int* count)
{
    return guarded([&] {
        require(count != nullptr, "null count");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        *count = n;
    });
}

int mlhip_ctx_create(int device_id, mlhip_ctx** out)
{
    return guarded([&] {
        require(out != nullptr, "null out");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            throw NoDevice("no HIP device available: this library has no CPU fallback (needs an AMD GPU, built for gfx950)");
        if (device_id < 0) device_id = env_int("MLHIP_DEVICE", env_int("LOCAL_RANK", 0));
        if (device_id >= n) device_id = device_id % n;
        auto* ctx = new mlhip_ctx;
        try {
            ctx->device = device_id;
            ctx->use();
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
            ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        } catch (...) {
            delete ctx;
            throw;
        }
        *out = ctx;
    });
}

int mlhip_ctx_destroy(mlhip_ctx* ctx)
{
    return guarded([&] {
        if (!ctx) return;
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->comm) drop_rccl(ctx);
        ctx->small_dev.release();
        ctx->small_host.release();
        for (int b = 0; b < 2; ++b) { ctx->up_stage[b].release(); ctx->up_pin[b].release(); }
        for (auto& p : ctx->pending) ctx->spare_events.push_back(p.second);
        for (auto& e : ctx->spare_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
    });
}

int mlhip_ctx_synchronize(mlhip_ctx* ctx)
{
    return guarded([&] { require(ctx, "null context"); ctx->use(); ctx->sync(); });
}
int mlhip_ctx_device(const mlhip_ctx* ctx, int* device_id)
{
    return guarded([&] { require(ctx && device_id, "null argument"); *device_id = ctx->device; });
}
int mlhip_ctx_stream(const mlhip_ctx* ctx, void** stream)
{
    return guarded([&] { require(ctx && stream, "null argument"); *stream = (void*)ctx->stream; });
}

int mlhip_ctx_set_allreduce(mlhip_ctx* ctx, mlhip_allreduce_fn fn, void* user, int on_device, int world_size, int rank)
{
    return guarded([&] {
        require(ctx, "null context");
        require(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size / rank");
        if (ctx->comm) drop_rccl(ctx);          // a caller-supplied hook replaces the library's own communicator
        ctx->reduce_fn = fn;
        ctx->reduce_user = user;
        ctx->reduce_on_device = on_device;
        ctx->world_size = fn ? world_size : 1;
        ctx->rank = fn ? rank : 0;
        // ranks sharing this host: LOCAL_WORLD_SIZE when a launcher (torchrun) exports it, else the whole world
        int local = ctx->world_size;
        if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
            const int v = std::atoi(e);
            if (v >= 1 && v <= ctx->world_size) local = v;
        }
        host::set_host_ranks(local);
    });
}

int mlhip_rccl_available(void)
{
    try { (void)Rccl::get(); return 1; } catch (...) { return 0; }
}

int mlhip_rccl_unique_id(void* unique_id)
{
    return guarded([&] {
        require(unique_id != nullptr, "null unique_id");
        static_assert(sizeof(ncclUniqueId) == MLHIP_RCCL_UNIQUE_ID_BYTES, "unique id size");
        const Rccl& r = Rccl::get();
        ncclUniqueId id;
        r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(unique_id, &id, sizeof id);
    });
}

int mlhip_ctx_init_rccl(mlhip_ctx* ctx, const void* unique_id, int world_size, int rank)
{
    return guarded([&] {
        require(ctx && unique_id, "null argument");
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        init_rccl(ctx, id, world_size, rank);
    });
}

int mlhip_ctx_init_rccl_file(mlhip_ctx* ctx, const char* path, int world_size, int rank)
{
    return guarded([&] {
        require(ctx && path && *path, "null argument");
        require(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size / rank");
        ncclUniqueId id;
        if (rank == 0) {
            const Rccl& r = Rccl::get();
            r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
            const std::string tmp = std::string(path) + ".tmp";     // written whole, then renamed: readers never see a part
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&id, 1, sizeof id, f) != sizeof id || std::fclose(f) != 0 || std::rename(tmp.c_str(), path) != 0)
                throw std::runtime_error(std::string("cannot write the RCCL rendezvous file ") + path);
        } else {
            const int limit_s = std::max(1, env_int("MLHIP_RCCL_TIMEOUT_S", 120));
            const int stale_s = std::max(1, env_int("MLHIP_RCCL_STALE_S", 600));
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                // the left-over file of an earlier job (one that died before rank 0 removed it) must not be taken for this job's
                struct stat st;
                const bool fresh = ::stat(path, &st) == 0 && std::time(nullptr) - st.st_mtime <= stale_s;
                if (FILE* f = fresh ? std::fopen(path, "rb") : nullptr) {
                    const size_t got = std::fread(&id, 1, sizeof id, f);
                    std::fclose(f);
                    if (got == sizeof id) break;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limit_s))
                    throw std::runtime_error(std::string("timed out waiting for the RCCL rendezvous file ") + path);
                usleep(20000);
            }
        }
        init_rccl(ctx, id, world_size, rank);
        // ncclCommInitRank returns when every rank has joined: the file has served; a rerun with the same path starts clean
        if (rank == 0) std::remove(path);
    });
}

int mlhip_ctx_rccl_ranks(const mlhip_ctx* ctx, int* nranks)
{
    return guarded([&] {
        require(ctx && nranks, "null argument");
        *nranks = 0;
        if (!ctx->comm) return;
        const Rccl& r = Rccl::get();
        r.check(r.CommCount(ctx->comm, nranks), "ncclCommCount");
    });
}

int mlhip_ctx_finalize_rccl(mlhip_ctx* ctx)
{
    return guarded([&] { require(ctx, "null context"); ctx->use(); drop_rccl(ctx); });
}

int mlhip_ctx_allreduce(mlhip_ctx* ctx, double* buf, size_t count)
{
    return guarded([&] {
        require(ctx && (buf || count == 0), "null argument");
        ctx->use();
        ctx->allreduce_host(buf, count);
    });
}
int mlhip_ctx_world(const mlhip_ctx* ctx, int* world_size, int* rank)
{
    return guarded([&] {
        require(ctx, "null context");
        if (world_size) *world_size = ctx->world_size;
        if (rank) *rank = ctx->rank;
    });
}

int mlhip_data_upload(mlhip_ctx* ctx, const double* x, uint32_t d, uint64_t n, int64_t ld, mlhip_data** out)
{
    return guarded([&] { require(out, "null out"); *out = upload_common(ctx, x, false, d, n, ld); });
}
int mlhip_data_upload_dev(mlhip_ctx* ctx, const double* x_dev, uint32_t d, uint64_t n, int64_t ld, mlhip_data** out)
{
    return guarded([&] { require(out, "null out"); *out = upload_common(ctx, x_dev, true, d, n, ld); });
}
int mlhip_data_free(mlhip_data* data)
{
    return guarded([&] {
        if (!data) return;
        (void)hipSetDevice(data->ctx->device);
        (void)hipStreamSynchronize(data->ctx->stream);
        delete data;
    });
}
int mlhip_data_shape(const mlhip_data* data, uint32_t* d, uint64_t* n_local, uint64_t* n_global)
{
    return guarded([&] {
        require(data, "null data");
        if (d) *d = (uint32_t)data->d;
        if (n_local) *n_local = data->n;
        if (n_global) *n_global = data->n_global;
    });
}
int mlhip_data_shift(const mlhip_data* data, double* shift)
{
    return guarded([&] {
        require(data && shift, "null argument");
        std::memcpy(shift, data->shift.data(), sizeof(double) * data->d);
    });
}

int mlhip_em_expectation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* mixing, const double* means,
                         const double* covariances, double* log_likelihood)
{
    return guarded([&] {
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

int mlhip_em_responsibilities(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* resp, int64_t ldr)
{
    return guarded([&] {
        check_em_args(ctx, data, K);
        require(resp || data->n == 0, "null argument");
        require(ldr >= (int64_t)data->n, "ldr must be >= n_local");
        require(data->have_estep && data->em_K == (int)K, "no E-step results on the device for this K");
        ensure_lw(data, (int)K);
        data->resp_dev.reserve(sizeof(double) * data->ldr * K);
        RespArgs a{data->lw.as<double>(), data->ldr, data->lse.as<double>(), data->n, (int)K,
                   data->resp_dev.as<double>(), data->ldr, nullptr};
        launch_em_responsibilities(a, ctx->stream);
        HIP_CHECK(hipGetLastError());
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(resp), sizeof(double) * ldr, data->resp_dev.as<char>(),
                         sizeof(double) * data->ldr, sizeof(double) * data->n, K);
    });
}

int mlhip_em_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint32_t* labels)
{
    return guarded([&] {
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

int mlhip_sample_covariance(mlhip_ctx* ctx, mlhip_data* data, double* mean, double* covariance)
{
    return guarded([&] {
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

int mlhip_kmeans_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* inertia,
                      uint64_t* n_changed, double* counts, double* centroids_out)
{
    return guarded([&] {
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
        check_em_args(ctx, data, 1);
        require(dist2 || data->n == 0, "null argument");
        require(data->km_have_old, "no K-means assignment on the device yet");
        ctx->sync();
        download_columns(ctx, reinterpret_cast<char*>(dist2), 0, data->km_mind.as<char>(), 0, sizeof(double) * data->n, 1);
    });
}

int mlhip_min_squared_distances(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* dist2)
{
    return guarded([&] {
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

int mlhip_random_partition_means(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* order, const uint32_t* offsets,
                                 double* means, double* sizes)
{
    return guarded([&] {
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

int mlhip_em_plan(const mlhip_data* data, uint32_t K, uint32_t* flags)
{
    return guarded([&] {
        require(data && flags && K >= 1, "null argument");
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

int mlhip_timing_enable(mlhip_ctx* ctx, int on)
{
    return guarded([&] {
        require(ctx, "null context");
        ctx->use();
        if (!on) ctx->resolve_timers();
        ctx->timing = on != 0;
    });
}
int mlhip_timing_reset(mlhip_ctx* ctx)
{
    return guarded([&] {
        require(ctx, "null context");
        ctx->use();
        ctx->resolve_timers();
        ctx->timers.clear();
    });
}
int mlhip_timing_get(mlhip_ctx* ctx, const char* name, double* avg_ms, uint64_t* launches)
{
    return guarded([&] {
        require(ctx && name && avg_ms && launches, "null argument");
        ctx->use();
        ctx->resolve_timers();
        auto it = ctx->timers.find(name);
        if (it == ctx->timers.end() || it->second.launches == 0) { *avg_ms = 0; *launches = 0; return; }
        *avg_ms = it->second.total_ms / (double)it->second.launches;
        *launches = it->second.launches;
    });
}

}  // extern "C"
