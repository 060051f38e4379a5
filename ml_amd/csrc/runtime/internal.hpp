// Internal to the runtime (runtime/*.cpp): the context and data objects behind the C ABI of include/mlhip.h, error mapping,
// and the functions the ABI entry points of the four families (context / data / EM / K-means) share. Not installed.
#pragma once
#include "mlhip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is dlopen'ed on first use (librccl is 570 MB; single-GPU users never pay for it)

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
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

namespace mlhip_rt {

extern thread_local std::string g_error;   // context.cpp

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

/// What a failed all-reduce hook becomes: a library-owned hook (RCCL, a device group's in-process sum) leaves its reason in the
/// calling thread's error text; a caller's hook only has its return code.
inline std::runtime_error hook_failure()
{
    return std::runtime_error(g_error.empty() ? std::string("all-reduce hook failed") : "all-reduce hook failed: " + g_error);
}

template <class F> int guarded(F&& f)
{
    g_error.clear();                   // (mlhip_last_error speaks of the LAST call; hook_failure reads what a hook left during this one)
    try { f(); return MLHIP_OK; }
    catch (const InvalidArgument& e) { g_error = e.what(); return MLHIP_E_INVALID_ARGUMENT; }
    catch (const DomainError& e) { g_error = e.what(); return MLHIP_E_DOMAIN; }
    catch (const NoDevice& e) { g_error = e.what(); return MLHIP_E_NO_DEVICE; }
    catch (const Unsupported& e) { g_error = e.what(); return MLHIP_E_UNSUPPORTED; }
    catch (const std::exception& e) { g_error = e.what(); return MLHIP_E_RUNTIME; }
}

/// Blocks a context's data handles have given back, kept for the next handle: a fit of the reference's benchmark size (N = 10k, d = 4,
/// K = 3) spends 1.3 of its 2.6 ms in hipMalloc / hipHostMalloc / hipFree (some 25 device and 7 pinned blocks per handle; hipFree
/// also synchronises the device). One pool per context = per stream: a block goes from one handle to the next in stream order, and
/// a handle is only released behind a stream synchronisation (mlhip_data_free), so no work of the previous owner is pending on it.
/// Blocks keep their exact allocation size and are handed out only for requests of at most that size and more than half of it
/// (powers of two up to 1 MB, multiples of 1 MB above), up to a capped total; the rest goes back to the driver.
/// MLHIP_POOL=0: every buffer straight from / to the driver.
struct BufferPool {
    static constexpr size_t kMaxBlock = size_t(64) << 20, kMaxDevice = size_t(512) << 20, kMaxPinned = size_t(64) << 20;
    std::mutex m;                                   // (a group's worker thread allocates, its caller's thread releases)
    std::vector<std::pair<size_t, void*>> dev, pin;
    size_t dev_bytes = 0, pin_bytes = 0;
    bool enabled = [] { const char* e = std::getenv("MLHIP_POOL"); return !(e && e[0] == '0'); }();

    static size_t block_size(size_t b)
    {
        if (b <= 256) return 256;
        if (b <= (size_t(1) << 20)) { size_t s = 256; while (s < b) s <<= 1; return s; }
        return (b + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    }
    void* take(bool pinned, size_t cap)
    {
        std::lock_guard<std::mutex> lock(m);
        auto& v = pinned ? pin : dev;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].first == cap) {
                void* p = v[i].second;
                v[i] = v.back(); v.pop_back();
                (pinned ? pin_bytes : dev_bytes) -= cap;
                return p;
            }
        return nullptr;
    }
    /// false: not kept (the caller frees it).
    bool give(bool pinned, size_t cap, void* p)
    {
        if (!enabled || cap > kMaxBlock) return false;
        std::lock_guard<std::mutex> lock(m);
        size_t& total = pinned ? pin_bytes : dev_bytes;
        if (total + cap > (pinned ? kMaxPinned : kMaxDevice)) return false;
        (pinned ? pin : dev).emplace_back(cap, p);
        total += cap;
        return true;
    }
    void drain()
    {
        std::lock_guard<std::mutex> lock(m);
        for (auto& b : dev) (void)hipFree(b.second);
        for (auto& b : pin) (void)hipHostFree(b.second);
        dev.clear(); pin.clear(); dev_bytes = pin_bytes = 0;
    }
};

/// Growable device buffer. `bytes` is what was asked for (the largest request so far: sizes derived from it -- grids, capacities --
/// do not depend on which block the pool handed out), `cap` the block behind it.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0, cap = 0;
    BufferPool* pool = nullptr;
    void reserve(size_t b)
    {
        if (b <= bytes) return;
        if (b <= cap) { bytes = b; return; }
        release();
        const size_t want = pool && pool->enabled ? BufferPool::block_size(b) : b;
        if (pool && pool->enabled) p = pool->take(false, want);
        if (!p) HIP_CHECK(hipMalloc(&p, want));
        bytes = b; cap = want;
    }
    void release()
    {
        if (p && !(pool && pool->give(false, cap, p))) (void)hipFree(p);
        p = nullptr; bytes = 0; cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0, cap = 0;
    BufferPool* pool = nullptr;
    void reserve(size_t b)
    {
        if (b <= bytes) return;
        if (b <= cap) { bytes = b; return; }
        release();
        const size_t want = pool && pool->enabled ? BufferPool::block_size(b) : b;
        if (pool && pool->enabled) p = pool->take(true, want);
        if (!p) HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        bytes = b; cap = want;
    }
    void release()
    {
        if (p && !(pool && pool->give(true, cap, p))) (void)hipHostFree(p);
        p = nullptr; bytes = 0; cap = 0;
    }
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
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;      // (only the device group needs it)
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // (optional: a device group cancels the collectives of a failed task)
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
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
            x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(sym("ncclCommAbort"));
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

}  // namespace mlhip_rt
using namespace mlhip_rt;   // (internal header: the runtime's own translation units only)

struct mlhip_group;   // group.cpp

struct mlhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    // Device group (mlhip_ctx_create_group): `group` is set on the context the caller holds -- it owns no stream of its own, every
    // entry point fans out to the group's shard contexts; `member_of` / `shard` are set on those.
    mlhip_group* group = nullptr;
    mlhip_group* member_of = nullptr;
    int shard = 0;
    size_t stage_bytes = size_t(128) << 20;   // upload staging chunk (two pinned + two device buffers of this size)
    // all-reduce hook
    mlhip_allreduce_fn reduce_fn = nullptr;
    void* reduce_user = nullptr;
    int reduce_on_device = 0, world_size = 1, rank = 0;
    ncclComm_t comm = nullptr;   // library-owned RCCL communicator (mlhip_ctx_init_rccl); its all-reduce is the hook then
    BufferPool pool;         // blocks of released data handles, for the next one
    // data handles alive on this context: destroying the context DETACHES them (their buffers stop referring to the pool, their context
    // pointer goes null), so a handle freed after its context releases its memory to the driver instead of touching freed state
    std::mutex handles_m;
    std::vector<mlhip_data*> handles;
    void adopt(mlhip_data* h) { std::lock_guard<std::mutex> lock(handles_m); handles.push_back(h); }
    void disown(mlhip_data* h)
    {
        std::lock_guard<std::mutex> lock(handles_m);
        handles.erase(std::remove(handles.begin(), handles.end(), h), handles.end());
    }
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
        g_error.clear();
        timed("allreduce", [&] { rc = reduce_fn(reduce_user, buf, count, 1, stream); });
        if (rc != 0) throw hook_failure();
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
            if (reduce_fn(reduce_user, v, count, 0, stream) != 0) throw hook_failure();
        }
    }
};

struct mlhip_data {
    mlhip_ctx* ctx = nullptr;
    // A block uploaded through a device group: `parts[s]` is shard s's resident block (rows first_row[s] .. first_row[s+1] of the
    // caller's sample); nothing else below is used then.
    std::vector<mlhip_data*> parts;
    std::vector<uint64_t> first_row;
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
    // mlhip_em_iterate (em_loop.cpp): parameters and the E-steps' records stay on the device between iterations, in a ring of three --
    // iteration i reads the records of slot i % 3 (params_dev / params_next / params_prev take turns) and writes pack and records
    // (i + 1) % 3; it_pack: [info (1 + 2K) | mixing (K) | means (K d) | covariances]; one pinned read-back slot and event per pack
    DevBuf params_next, params_prev, it_pack[3];
    DevBuf it_sync, it_xch;       // device-resident loop (em_resident.hip): arrival counter / give-up flag, exchange blocks
    DevBuf close_work;            // closing arithmetic at d > 64 (em_close_big.hip): the components' L and W
    PinnedBuf it_info_slot[3], it_history;   // it_history: [result words | log-likelihood history] of the resident loop
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
    DevBuf kpp_w, kpp_scr;               // mlhip_kpp_draw: the running-minimum weights, block sums / offsets / result
    DevBuf km_ticket;                    // kmeans_reduce_close_kernel: the arrival counter, never reset ...
    unsigned km_ticket_base = 0;         // ... and the tickets drawn from it so far (wraps with it)
    int km_cur = 0;
    bool km_have_old = false;

    /// Every buffer of the handle draws from / returns to the context's pool.
    void attach_pool(BufferPool* pool)
    {
        for (DevBuf* b : {&xt, &shift_dev, &lw, &lse, &esum, &ll_partials, &params_dev, &partials, &stats_dev, &resp_dev,
                          &labels_dev, &km_labels[0], &km_labels[1], &km_cent, &km_cent_next, &km_partials, &km_out, &km_mind, &km_probe, &km_scale, &km_cnorm, &km_xt_pad, &kpp_w, &kpp_scr, &km_ticket,
                          &refine_shift, &refine_stats, &params_next, &params_prev, &it_pack[0], &it_pack[1], &it_pack[2], &it_sync, &it_xch, &close_work})
            b->pool = pool;
        for (PinnedBuf* b : {&params_host, &stats_host, &km_host, &it_info_slot[0], &it_info_slot[1], &it_info_slot[2], &it_history}) b->pool = pool;
    }

    ~mlhip_data()
    {
        for (mlhip_data* p : parts) mlhip_data_free(p);
        for (DevBuf* b : {&xt, &shift_dev, &lw, &lse, &esum, &ll_partials, &params_dev, &partials, &stats_dev, &resp_dev,
                          &labels_dev, &km_labels[0], &km_labels[1], &km_cent, &km_cent_next, &km_partials, &km_out, &km_mind, &km_probe, &km_scale, &km_cnorm, &km_xt_pad, &kpp_w, &kpp_scr, &km_ticket,
                          &refine_shift, &refine_stats, &params_next, &params_prev, &it_pack[0], &it_pack[1], &it_pack[2], &it_sync, &it_xch, &close_work})
            b->release();
        for (auto& sl : it_info_slot) sl.release();
        for (auto& e : it_event) if (e) (void)hipEventDestroy(e);
        params_host.release(); stats_host.release(); km_host.release(); it_history.release();
    }
};

namespace mlhip_rt {

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

inline void require(bool ok, const char* msg) { if (!ok) throw InvalidArgument(msg); }

/// What one K-means pass runs on: the data block (a zero-padded copy where the matrix-core kernel needs one) and its rows.
struct KmBlock {
    const double* xt;
    int D;
};

// ---- shared between the families (definitions: context.cpp, data.cpp, em.cpp, kmeans.cpp) ----

int env_int(const char* name, int fallback);

/// memcpy split over a few threads: one core moves ~10 GB/s (less into untouched pages), below the PCIe rate it feeds.
void copy_bytes(void* dst, const void* src, size_t bytes);

void finish_upload(mlhip_data* dt);

mlhip_data* upload_common(mlhip_ctx* ctx, const double* x, bool on_device, uint32_t d, uint64_t n, int64_t ld);

/// Device -> pageable host copy of `cols` columns of `col_bytes` bytes each (source / destination pitches given), staged through
/// the context's two pinned buffers: the CPU unpacks chunk i while the DMA engine fetches chunk i+1. A direct copy into
/// pageable memory runs at ~3 GB/s on this platform; this one at PCIe rate.
void download_columns(mlhip_ctx* ctx, char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t col_bytes, size_t cols);

void ensure_em_workspace(mlhip_data* dt, int K);

/// Builds the per-component records for the E-step kernel that fits (d, env) and uploads them to params_dev.
void prepare_estep(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs, DevBuf* target = nullptr);

/// E-step kernel on the records in params_dev: fills lw and -- unless the statistics kernel is going to normalise the
/// log-responsibilities itself (`with_lse` false, matrix-core kernel only) -- lse and the log-likelihood partials.
/// `records` / `fold`: another record buffer than params_dev and its form (mlhip_em_iterate keeps a ring of them); default: params_dev
/// and dt->estep_fold.
void launch_estep(mlhip_data* dt, int K, bool with_lse = true, const DevBuf* records = nullptr, int fold = -1);

void run_estep(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs, bool with_lse = true);

/// After a fused step only lse exists on the device; whoever needs the log-responsibility block (labels,
/// responsibilities, a separate M-step, the refinement pass) gets it rebuilt from the same parameter records.
void ensure_lw(mlhip_data* dt, int K);

/// All-reduces the reduced statistics buffer [K*F stats, ll_sum] and leaves it in stats_host.
void collect_stats(mlhip_data* dt, int K, size_t count = 0);

/// One EM iteration's device work in a single kernel where the shape allows (d <= 6, K <= 32 or d <= 4, K <= 64: em_fused_small.hip): no
/// N x K block in HBM. MLHIP_FUSED=0 keeps the two-kernel path. Returns false when the shape is not covered.
bool fused_step_applies(const mlhip_data* dt, int K);

/// The fused kernel + reduction on the records already in params_dev; statistics end in stats_dev (and, with `collect`, all-
/// reduced in stats_host).
void launch_fused_step(mlhip_data* dt, int K, bool collect, const DevBuf* records = nullptr);

bool run_fused_step(mlhip_data* dt, int K, const double* mixing, const double* means, const double* covs);

/// Runs the statistics kernel on log-responsibilities (mode kFromLogResp: the E-step's lw/lse) or on plain
/// responsibilities `resp_dev` ([K][ld_resp], ld_resp >= n_pad), all-reduces, leaves [K*F stats, ll_sum] in stats_host.
void run_mstats(mlhip_data* dt, int K, int mode, const double* resp_dev, size_t ld_resp, bool with_ll, bool collect = true);

double log_two_pi();

double ll_from_stats(const mlhip_data* dt, int K);

void check_em_args(mlhip_ctx* ctx, mlhip_data* dt, uint32_t K);

/// Ratio (mean offset from the shared shift)^2 / variance above which a component's covariance is recomputed about its
/// own mean. The one-GEMM statistics share one shift (the global mean), so Sigma_k = M2'/S0 - m m^T cancels
/// ~log10(ratio) digits: measured relative error ~3e-15 * ratio. 1e4 keeps every covariance within ~3e-11 of the
/// two-pass form the reference uses (ML/EM.cpp:245-250). MLHIP_REFINE_RATIO overrides; <= 0 disables the refinement.
double refine_ratio();

/// Second statistics pass for ONE component with the shift at that component's new mean (K = 1 launch of the same
/// kernels on column k of the responsibilities of the last pass), all-reduced like the first; replaces covariance k
/// (and adds the tiny mean correction). Tight clusters far from the global mean need it; the headline shapes never do.
void refine_component(mlhip_data* dt, int k, double* mean_k, double* cov_k);

void finalize_out(mlhip_data* dt, int K, double* mixing_out, double* means_out, double* cov_out);

/// One diagonal-covariance EM iteration's device work (em_diag.hip) with the statistics shift at `shift_dev`; leaves the
/// all-reduced [K * (2d+1) statistics, ll_sum] in stats_host. The records must already be in params_dev.
void run_diag_kernel(mlhip_data* dt, int K, const double* shift_dev, bool collect = true, const DevBuf* records = nullptr);

void ensure_km_workspace(mlhip_data* dt, int K);

KmBlock km_block(mlhip_data* dt, int K);

/// Host centroids [K][d] -> the device table km_cent [K][D] (padded coordinates zero).
void km_upload_centroids(mlhip_data* dt, int K, const KmBlock& b, const double* centroids);

/// Assignment (+ optional accumulation) against the table in km_cent, partials reduced into km_out =
/// [inertia, changed, counts, sums] and summed across ranks there when the all-reduce works on device memory.
/// `close_next` (with accumulate, single rank): the closing arithmetic rides in the reduction's launch (kmeans_reduce_close_kernel) --
/// means into km_out, the next table into close_next, the block into the pinned close_mirror (may be null); returns true when it did.
bool km_launch(mlhip_data* dt, int K, const KmBlock& b, bool accumulate, double* min_dist_out, double* close_next = nullptr,
               double* close_mirror = nullptr);

/// km_out -> km_host (`count` doubles), summed across ranks on the host when the all-reduce works on host memory.
void km_fetch(mlhip_data* dt, size_t count);

/// Assignment (+ optional accumulation); leaves all-reduced [inertia, changed, counts, sums] in km_host.
void run_kmeans(mlhip_data* dt, int K, const double* centroids, bool accumulate, double* min_dist_out = nullptr);

/// update_step's closing arithmetic on the host (ML/KMeans.cpp:180-192 as sums / counts; empty cluster -> origin, :184).
void km_close_host(const double* r, int K, int d, double* counts, double* centroids_out);

/// The step loop of KMeans::fit_once (ML/KMeans.cpp:80-110). With the all-reduce on device memory (or none) the centroid
/// table never leaves the device between trips: sums -> means -> next table by launch_kmeans_close, one read-back per trip
/// for the two stopping tests. With a host-memory all-reduce (gloo rehearsals) every trip goes through run_kmeans.
void km_iterate(mlhip_data* dt, int K, double* centroids, double* old_centroids, uint32_t max_steps, double atol,
                uint32_t* steps_done, int* converged, double* inertia, double* counts);

/// K within one row-block group of the wide statistics kernel: the matrix-core E-step writes the log-responsibilities only and
/// the statistics kernel normalises them (one exp per pair in the iteration); otherwise the E-step keeps its online
/// log-sum-exp. MLHIP_SELF_NORM=0 forces the latter (A/B runs).
bool self_norm_applies(const mlhip_data* dt, int K);

/// One full-covariance EM iteration with the closing arithmetic on the HOST (the body of mlhip_em_step).
void em_step_full(mlhip_data* data, int K, const double* mixing, const double* means, const double* covariances,
                  double* log_likelihood, double* mixing_out, double* means_out, double* covariances_out);

/// Sums `count` doubles at the head of stats_dev across ranks, whatever kind of hook is installed (device buffer on the
/// stream, or a host buffer: down, hook, up). No-op on a single rank.
void allreduce_stats_dev(mlhip_data* dt, size_t count);

/// Records of a diagonal-covariance parameter set -> `target` (padded to whole 16-component row blocks with neutral records).
void upload_diag_records(mlhip_data* data, int K, const double* mixing, const double* means, const double* variances, DevBuf& target);

/// Same cancellation guard as the full-covariance path (refine_ratio): a component whose mean sits far from the shared shift,
/// measured in its own standard deviations, gets its variances from a second pass with the shift at its new mean (the E part of
/// that pass re-evaluates the SAME input parameters, still in params_dev).
void refine_diag(mlhip_data* data, int K, const double* mixing_out, double* means_out, double* variances_out);

void em_step_diag(mlhip_data* data, int K, const double* mixing, const double* means, const double* variances,
                  double* log_likelihood, double* mixing_out, double* means_out, double* variances_out);

/// The loop of EM::fit (ML/EM.cpp:143-170) with everything between two convergence tests on the device (em_loop.cpp): E-step,
/// statistics, all-reduce, closing arithmetic + next records (em_close.hip); per iteration the host reads back 1 + 2K doubles (log-
/// likelihood sum, refinement flags, FOLD criterion) and decides. A flagged component (far, tight cluster) sends that one
/// iteration through the host closing with its refinement pass, exactly as mlhip_em_step would. MLHIP_DEVICE_CLOSE=0, or
/// d > 64, runs the whole loop through the per-step functions.
void em_iterate(mlhip_data* data, int K, bool diag, double* mixing, double* means, double* covs, uint32_t max_steps, double atol,
                double rtol, uint32_t* steps_done, int* converged, double* log_likelihood, double* history);

/// The all-reduce hook of a context that owns an RCCL communicator: one ncclAllReduce(double, sum), in place, on the
/// context's stream -- ordered with the kernels before it and the copies after it, no host synchronisation.
int rccl_allreduce_hook(void* user, double* buf, size_t count, int on_device, void* stream);

void drop_rccl(mlhip_ctx* ctx);

void init_rccl(mlhip_ctx* ctx, const ncclUniqueId& id, int world_size, int rank);

/// A failed C-ABI status (with the calling thread's message) as the exception `guarded` maps back to it: the device group calls
/// the entry points of its shards on worker threads and hands their failures to the thread of the caller.
[[noreturn]] void throw_status(int status, const std::string& message);
inline void check_status(int status) { if (status != MLHIP_OK) throw_status(status, g_error); }

/// mlhip_ctx_create for one device, without the environment defaults (context.cpp).
mlhip_ctx* create_single_context(int device_id);
void destroy_single_context(mlhip_ctx* ctx);

// ---- device group (group.cpp): one caller-visible context over n shards. Every function is the group form of the entry point of
// the same name: host arrays are the caller's WHOLE arrays (rows of all shards), results are what a single context would return.
namespace grp {
mlhip_ctx* create(int n_shards, const int* device_ids);
void destroy(mlhip_ctx* ctx);
void synchronize(mlhip_ctx* ctx);
int shard_count(const mlhip_ctx* ctx);
mlhip_ctx* shard_context(const mlhip_ctx* ctx, int shard);
const char* reduce_kind(const mlhip_ctx* ctx);
mlhip_data* upload(mlhip_ctx* ctx, const double* x, bool on_device, uint32_t d, uint64_t n, int64_t ld);
void sample_covariance(mlhip_ctx* ctx, mlhip_data* data, double* mean, double* covariance);
void xxt_xy(mlhip_ctx* ctx, mlhip_data* data, const double* y, double* xxt, double* xy);
void random_partition_means(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* order, const uint32_t* offsets, double* means,
                            double* sizes);
void em_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, bool diag, const double* mixing, const double* means, const double* covs,
             double* log_likelihood, double* mixing_out, double* means_out, double* covs_out);
void em_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, int covariance_type, double* mixing, double* means, double* covs,
                uint32_t max_steps, double atol, double rtol, uint32_t* steps_done, int* converged, double* log_likelihood,
                double* history);
void em_expectation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* mixing, const double* means, const double* covs,
                    double* log_likelihood);
/// source: 0 = the last E-step (mlhip_em_maximisation), 1 = caller's responsibilities, 2 = caller's labels
void em_maximisation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, int source, const double* resp, int64_t ldr, const uint32_t* labels,
                     double* mixing_out, double* means_out, double* covs_out);
void em_responsibilities(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* resp, int64_t ldr, uint64_t first, uint64_t count);
void em_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint32_t* labels);
/// accumulate: the update's sums as well (mlhip_kmeans_step), else the assignment only
void kmeans_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, bool accumulate, const double* centroids, double* inertia,
                 uint64_t* n_changed, double* counts, double* centroids_out);
void kmeans_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* centroids, double* old_centroids, uint32_t max_steps,
                    double atol, uint32_t* steps_done, int* converged, double* inertia, double* counts);
void kmeans_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t* labels);
void kmeans_distances(mlhip_ctx* ctx, mlhip_data* data, double* dist2);
void kpp_draw(mlhip_ctx* ctx, mlhip_data* data, const double* centroid, int first, double u, uint64_t first_row, uint64_t* index,
              int* certain, double* weights_out);
void kpp_weights(mlhip_ctx* ctx, mlhip_data* data, double* weights_out);
void min_squared_distances(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* dist2);
void timing_enable(mlhip_ctx* ctx, int on);
void timing_reset(mlhip_ctx* ctx);
void timing_get(mlhip_ctx* ctx, const char* name, double* avg_ms, uint64_t* launches);
}  // namespace grp

}  // namespace mlhip_rt
