// Device group: ONE caller-visible context over n shards (`mlhip_ctx_create(n_devices, device_ids)`, "one host
// thread drives all devices"), so that the reference's one-process API -- bool EM::fit(Eigen::Ref<const MatrixXd>) (ML/EM.cpp:91),
// KMeans::fit (ML/KMeans.cpp:25) -- reaches every GPU of the node without a launcher. The caller's d x N block is row-sharded over
// the shards at upload; every entry point of mlhip.h called with the group's context fans out to the shards' ordinary contexts, one
// persistent host thread per shard, and the shards' statistics meet in the per-iteration all-reduce exactly as the ranks of a
// multi-process job do (the shard contexts carry world_size = n, rank = shard, and run the unchanged per-context code):
//   * shards on distinct GPUs: one RCCL communicator per shard from ncclCommInitAll, ncclAllReduce on the shard's stream;
//   * otherwise (several shards on one GPU -- how a one-GPU box rehearses the 8-GPU configurations at full size -- or
//     MLHIP_GROUP_REDUCE=direct): an in-process all-reduce -- every shard copies its buffer into its own slot, waits (on its stream) for
//     the other shards' slots, and sums all slots in shard order with the same kernel: bit-identical on every shard, reproducible from
//     run to run, no library involved (slots on other GPUs are read through peer access over xGMI).
#include "internal.hpp"
#include "shard_team.hpp"

struct mlhip_group {
    int n = 0;
    std::vector<mlhip_ctx*> shard;      // owned: ordinary contexts with world_size = n, rank = shard index
    std::vector<int> devices;
    enum Reduce { kNone, kRccl, kDirect } reduce = kNone;
    std::string rccl_error;             // why an automatic choice fell back from RCCL to the in-process sum (empty: it did not)
    std::string direct_kind_text;       // "group-direct (RCCL unavailable: ...)" for mlhip_ctx_reduce_kind
    ShardTeam team;                     // the shard threads, their barrier, the failure protocol (shard_team.hpp)
    std::atomic<bool> dead{false};      // RCCL mode, after a failure: the communicators were aborted, the group can only be closed
    // ---- in-process all-reduce (kDirect): the protocol is SlotAllreduce (shard_team.hpp); these are the device objects it moves --
    // two generations of one slot per shard, an event pair per slot
    struct DeviceOps;
    std::vector<DevBuf> slot[2];
    std::vector<hipEvent_t> ready[2], consumed[2];
    mlhip_rt::SlotAllreduce<DeviceOps> exchange;
    // ---- test hooks, read ONCE when the group is created (tests/test_gpu_group.py): MLHIP_GROUP_TEST_PERTURB=s -- shard s's copy of
    // a statistics sum differs in its last digits (the checksum exchange must fail the fit); MLHIP_GROUP_TEST_FAIL=s:k -- shard s
    // fails, alone, in its k-th all-reduce (the others must come back with its error instead of waiting for it)
    int test_perturb_shard = -1, test_fail_shard = -1;
    uint64_t test_fail_at = 0;
    std::vector<uint64_t> allreduces;   // per shard, counted for the hook above
};

namespace mlhip_rt {
namespace {

}  // namespace
}  // namespace mlhip_rt

/// SlotAllreduce's device operations: shard r's stream, the slots' buffers and events. (Slots on other GPUs are read through peer access.)
struct mlhip_group::DeviceOps {
    mlhip_group* g;
    hipStream_t stream(int r) const { return g->shard[(size_t)r]->stream; }
    void sync_stream(int r) { HIP_CHECK(hipStreamSynchronize(stream(r))); }
    void reserve_slots(int r, size_t doubles) { for (int p = 0; p < 2; ++p) g->slot[p][(size_t)r].reserve(sizeof(double) * doubles); }
    void release_slots(int r) { for (int p = 0; p < 2; ++p) g->slot[p][(size_t)r].release(); }
    void wait_consumed(int r, int p, int q) { HIP_CHECK(hipStreamWaitEvent(stream(r), g->consumed[p][(size_t)q], 0)); }
    void publish(int r, int p, const double* buf, size_t count)
    {
        HIP_CHECK(hipMemcpyAsync(g->slot[p][(size_t)r].p, buf, sizeof(double) * count, hipMemcpyDeviceToDevice, stream(r)));
        HIP_CHECK(hipEventRecord(g->ready[p][(size_t)r], stream(r)));
    }
    void wait_ready(int r, int p, int q) { HIP_CHECK(hipStreamWaitEvent(stream(r), g->ready[p][(size_t)q], 0)); }
    void sum(int r, int p, double* buf, size_t count)
    {
        GroupSumSlots slots{};
        for (int q = 0; q < g->n; ++q) slots.p[q] = g->slot[p][(size_t)q].as<double>();
        launch_group_sum(slots, g->n, buf, count, stream(r));
        HIP_CHECK(hipGetLastError());
    }
    void mark_consumed(int r, int p) { HIP_CHECK(hipEventRecord(g->consumed[p][(size_t)r], stream(r))); }
};

namespace mlhip_rt {
namespace {

/// After a failed task the shards' streams may hold half an all-reduce and -- when a slot growth was interrupted -- the shards may
/// disagree on the slots' capacity: SlotAllreduce::recover (every shard drains its stream, all meet, every shard drops its slots,
/// all meet again).
void recover_shard(mlhip_group* g, int s)
{
    mlhip_ctx* c = g->shard[(size_t)s];
    (void)hipSetDevice(c->device);
    mlhip_group::DeviceOps ops{g};
    if (g->exchange.active()) {
        g->exchange.recover(ops, g->team.barrier, s);
    } else {
        (void)hipStreamSynchronize(c->stream);
        g->team.barrier.wait();
    }
}

/// A shard failed on its own (out of memory, a HIP error): the others may be INSIDE ncclAllReduce + hipStreamSynchronize waiting for
/// it, where no barrier of ours can reach them. RCCL mode: abort every communicator -- the blocked streams return with an error --
/// and mark the group dead (aborted communicators cannot be used again; the caller creates a new group).
void cancel_collectives(mlhip_group* g)
{
    if (g->reduce != mlhip_group::kRccl) return;
    g->dead.store(true);
    try {
        const Rccl& r = Rccl::get();
        if (!r.CommAbort) return;
        for (mlhip_ctx* c : g->shard)
            if (c->comm) (void)r.CommAbort(c->comm);
    } catch (...) {
    }
}

void run_on_shards(mlhip_group* g, const std::function<void(int)>& f)
{
    if (g->dead.load())
        throw std::runtime_error("this device group failed inside an RCCL collective and its communicators were aborted: close it "
                                 "(mlhip_ctx_destroy) and create a new one");
    g->team.run(f);
}

/// The all-reduce hook of a shard in kDirect mode (see the head of this file). Every shard calls it with the same count, in the same
/// order, each from its own thread with its own device current.
int direct_allreduce_hook(void* user, double* buf, size_t count, int on_device, void* stream_)
{
    auto* c = static_cast<mlhip_ctx*>(user);
    mlhip_group* g = c ? c->member_of : nullptr;
    if (!g || !on_device || static_cast<hipStream_t>(stream_) != c->stream) return 1;
    try {
        const int r = c->shard;
        if (g->test_fail_shard == r && ++g->allreduces[(size_t)r] == g->test_fail_at)
            throw std::runtime_error("injected failure of one shard (MLHIP_GROUP_TEST_FAIL)");
        mlhip_group::DeviceOps ops{g};
        g->exchange.allreduce(ops, g->team.barrier, r, buf, count);
        if (g->test_perturb_shard == r && count > 64) {
            // (tests only) this shard's copy of a statistics sum is made to differ in its last digits: the end-of-fit checksum
            // exchange of the shards must catch it (short vectors -- that exchange itself -- are left alone)
            double v = 0;
            HIP_CHECK(hipMemcpyAsync(&v, buf, sizeof v, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            v *= 1.0 + 1e-12;
            HIP_CHECK(hipMemcpyAsync(buf, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

mlhip_group* group_of(const mlhip_ctx* ctx)
{
    require(ctx && ctx->group, "not a device group context");
    return ctx->group;
}

void check_group_data(const mlhip_ctx* ctx, const mlhip_data* data)
{
    require(ctx && data, "null context or data");
    require(data->ctx == ctx, "data belongs to another context");
    require((int)data->parts.size() == ctx->group->n, "data was not uploaded through this device group");
}

template <class F> void each_shard(mlhip_ctx* ctx, F&& f)
{
    mlhip_group* g = group_of(ctx);
    run_on_shards(g, [&](int s) { f(s, g->shard[(size_t)s]); });
}

uint64_t rows_of(const mlhip_data* data, int s) { return data->first_row[(size_t)s + 1] - data->first_row[(size_t)s]; }

void enable_peer_access(const std::vector<int>& devices)
{
    for (int a : devices)
        for (int b : devices) {
            if (a == b) continue;
            int can = 0;
            HIP_CHECK(hipDeviceCanAccessPeer(&can, a, b));
            if (!can)
                throw Unsupported("device group: GPUs " + std::to_string(a) + " and " + std::to_string(b) +
                                  " cannot access each other's memory (the in-process all-reduce needs peer access; use RCCL: "
                                  "MLHIP_GROUP_REDUCE=rccl)");
            HIP_CHECK(hipSetDevice(a));
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
            (void)hipGetLastError();
        }
}

}  // namespace

[[noreturn]] void throw_status(int status, const std::string& message)
{
    switch (status) {
    case MLHIP_E_INVALID_ARGUMENT: throw InvalidArgument(message);
    case MLHIP_E_DOMAIN: throw DomainError(message);
    case MLHIP_E_NO_DEVICE: throw NoDevice(message);
    case MLHIP_E_UNSUPPORTED: throw Unsupported(message);
    default: throw std::runtime_error(message);
    }
}

namespace grp {

mlhip_ctx* create(int n_shards, const int* device_ids)
{
    require(n_shards >= 1 && n_shards <= kGroupMaxShards, "a device group has 1 to 64 shards");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw NoDevice("no HIP device available: this library has no CPU fallback (needs an AMD GPU, built for gfx950)");
    auto* g = new mlhip_group;
    auto* ctx = new mlhip_ctx;
    ctx->group = g;
    try {
        g->n = n_shards;
        bool distinct = true;
        for (int s = 0; s < n_shards; ++s) {
            const int dev = device_ids ? device_ids[s] : s % count;
            require(dev >= 0 && dev < count, "device group: no such GPU");
            for (int q : g->devices) distinct = distinct && q != dev;
            g->devices.push_back(dev);
        }
        for (int s = 0; s < n_shards; ++s) {
            mlhip_ctx* c = create_single_context(g->devices[(size_t)s]);
            g->shard.push_back(c);
            c->member_of = g;
            c->shard = s;
            c->stage_bytes = size_t(32) << 20;             // n x 4 staging buffers: keep the pinned footprint small
        }
        ctx->device = g->devices[0];
        ctx->num_cus = g->shard[0]->num_cus;
        if (n_shards > 1) {
            // how the shards' statistics meet: RCCL between distinct GPUs, the in-process sum otherwise
            const char* env = std::getenv("MLHIP_GROUP_REDUCE");
            const std::string want = env ? env : "";
            require(want.empty() || want == "rccl" || want == "direct", "MLHIP_GROUP_REDUCE must be rccl or direct");
            bool rccl = want == "rccl" || (want.empty() && distinct && mlhip_rccl_available());
            if (rccl && !distinct) throw Unsupported("device group: RCCL refuses two ranks on one GPU (use MLHIP_GROUP_REDUCE=direct)");
            if (rccl) {
                const Rccl& r = Rccl::get();
                if (!r.CommInitAll) throw std::runtime_error("librccl has no ncclCommInitAll");
                std::vector<ncclComm_t> comms((size_t)n_shards, nullptr);
                const ncclResult_t rc = r.CommInitAll(comms.data(), n_shards, g->devices.data());
                if (rc != ncclSuccess) {
                    if (want == "rccl") r.check(rc, "ncclCommInitAll");
                    rccl = false;                            // (auto: fall back to the in-process sum -- and say so, once)
                    g->rccl_error = std::string("ncclCommInitAll: ") + r.GetErrorString(rc);
                    g->direct_kind_text = "group-direct (RCCL unavailable: " + g->rccl_error + ")";
                    std::fprintf(stderr, "[mlhip] device group: RCCL between the %d GPUs is not available (%s); the shards' statistics are "
                                         "summed in-process over peer access instead (MLHIP_GROUP_REDUCE=rccl makes this an error)\n",
                                 n_shards, g->rccl_error.c_str());
                } else {
                    for (int s = 0; s < n_shards; ++s) {
                        mlhip_ctx* c = g->shard[(size_t)s];
                        c->comm = comms[(size_t)s];
                        c->reduce_fn = rccl_allreduce_hook;
                        c->reduce_user = c;
                    }
                    g->reduce = mlhip_group::kRccl;
                }
            }
            if (!rccl) {
                std::vector<int> unique;
                for (int dev : g->devices)
                    if (std::find(unique.begin(), unique.end(), dev) == unique.end()) unique.push_back(dev);
                enable_peer_access(unique);
                for (int p = 0; p < 2; ++p) {
                    g->slot[p].resize((size_t)n_shards);
                    g->ready[p].assign((size_t)n_shards, nullptr);
                    g->consumed[p].assign((size_t)n_shards, nullptr);
                }
                g->exchange.init(n_shards);
                for (int s = 0; s < n_shards; ++s) {
                    mlhip_ctx* c = g->shard[(size_t)s];
                    c->use();
                    for (int p = 0; p < 2; ++p) {
                        HIP_CHECK(hipEventCreateWithFlags(&g->ready[p][(size_t)s], hipEventDisableTiming));
                        HIP_CHECK(hipEventCreateWithFlags(&g->consumed[p][(size_t)s], hipEventDisableTiming));
                    }
                    c->reduce_fn = direct_allreduce_hook;
                    c->reduce_user = c;
                }
                g->reduce = mlhip_group::kDirect;
            }
            for (mlhip_ctx* c : g->shard) {
                c->reduce_on_device = 1;
                c->world_size = n_shards;
                c->rank = c->shard;
            }
            host::set_host_ranks(n_shards);                 // the shards' host threads share this machine's cores
        }
        // (test hooks: read once, here -- never on the path of an all-reduce)
        if (const char* e = std::getenv("MLHIP_GROUP_TEST_PERTURB")) g->test_perturb_shard = std::atoi(e);
        if (const char* e = std::getenv("MLHIP_GROUP_TEST_FAIL")) {
            g->test_fail_shard = std::atoi(e);
            const char* colon = std::strchr(e, ':');
            g->test_fail_at = colon ? std::strtoull(colon + 1, nullptr, 10) : 1;
        }
        g->allreduces.assign((size_t)n_shards, 0);
        g->team.start(n_shards, [g](int s) { recover_shard(g, s); }, [g](int) { cancel_collectives(g); });
    } catch (...) {
        destroy(ctx);
        throw;
    }
    return ctx;
}

void destroy(mlhip_ctx* ctx)
{
    if (!ctx) return;
    mlhip_group* g = ctx->group;
    if (g) {
        g->team.stop();
        if (g->n > 1) host::set_host_ranks(1);               // (the shards' share of the host cores was a property of this group)
        for (size_t s = 0; s < g->shard.size(); ++s) {
            mlhip_ctx* c = g->shard[s];
            (void)hipSetDevice(c->device);
            if (c->stream) (void)hipStreamSynchronize(c->stream);
        }
        for (size_t s = 0; s < g->shard.size(); ++s) {
            mlhip_ctx* c = g->shard[s];
            (void)hipSetDevice(c->device);
            for (int p = 0; p < 2; ++p) {
                if (s < g->slot[p].size()) g->slot[p][s].release();
                if (s < g->ready[p].size() && g->ready[p][s]) (void)hipEventDestroy(g->ready[p][s]);
                if (s < g->consumed[p].size() && g->consumed[p][s]) (void)hipEventDestroy(g->consumed[p][s]);
            }
            destroy_single_context(c);                       // (drops the shard's RCCL communicator too)
        }
        delete g;
    }
    {
        std::lock_guard<std::mutex> lock(ctx->handles_m);
        for (mlhip_data* h : ctx->handles) h->ctx = nullptr;   // (group blocks that outlive the group: their parts are detached already)
        ctx->handles.clear();
    }
    delete ctx;
}

void synchronize(mlhip_ctx* ctx)
{
    for (mlhip_ctx* c : group_of(ctx)->shard) { c->use(); c->sync(); }
}

int shard_count(const mlhip_ctx* ctx) { return ctx->group ? ctx->group->n : 1; }

mlhip_ctx* shard_context(const mlhip_ctx* ctx, int shard)
{
    mlhip_group* g = group_of(ctx);
    require(shard >= 0 && shard < g->n, "no such shard");
    return g->shard[(size_t)shard];
}

const char* reduce_kind(const mlhip_ctx* ctx)
{
    if (ctx->group) {
        switch (ctx->group->reduce) {
        case mlhip_group::kRccl: return "group-rccl";
        case mlhip_group::kDirect:
            // (an automatic choice that fell back from RCCL carries the reason: the bench line shows it)
            return ctx->group->rccl_error.empty() ? "group-direct" : ctx->group->direct_kind_text.c_str();
        default: return "none";
        }
    }
    if (!ctx->reduce_fn) return "none";
    if (ctx->reduce_fn == rccl_allreduce_hook) return "rccl";
    return ctx->reduce_on_device ? "hook-device" : "hook-host";
}

mlhip_data* upload(mlhip_ctx* ctx, const double* x, bool on_device, uint32_t d, uint64_t n, int64_t ld)
{
    mlhip_group* g = group_of(ctx);
    require(x != nullptr || n == 0, "null data");
    require(d >= 1, "At least one dimension required");
    require(ld >= (int64_t)d, "ld must be >= d");
    auto* gd = new mlhip_data;
    try {
        gd->ctx = ctx;
        gd->d = (int)d;
        gd->D = padded_dim((int)d);
        if (gd->D < 0) throw Unsupported("dimension d > 4096 is not supported");
        gd->n_global = n;
        gd->n = (uint32_t)std::min<uint64_t>(n, 0xffffffffull);
        gd->parts.assign((size_t)g->n, nullptr);
        // contiguous, balanced row shards that tile [0, n) -- the split of ml_amd.dist.shard_bounds, so that a group reproduces the
        // multi-process job of the same size
        gd->first_row.assign((size_t)g->n + 1, 0);
        const uint64_t base = n / (uint64_t)g->n, rem = n % (uint64_t)g->n;
        for (int s = 0; s < g->n; ++s) gd->first_row[(size_t)s + 1] = gd->first_row[(size_t)s] + base + ((uint64_t)s < rem ? 1 : 0);
        if (on_device && n) {
            // a block already in device memory must be readable from every shard's GPU: its own GPU, or a peer with access enabled
            // (a kernel that reads memory it cannot reach faults -- and a fault can take the node's GPUs down)
            hipPointerAttribute_t attr{};
            if (hipPointerGetAttributes(&attr, x) != hipSuccess) { (void)hipGetLastError(); throw InvalidArgument("x_dev is not a device pointer"); }
            std::vector<int> others;
            for (int dev : g->devices)
                if (dev != attr.device && std::find(others.begin(), others.end(), dev) == others.end()) others.push_back(dev);
            for (int dev : others) {
                int can = 0;
                HIP_CHECK(hipDeviceCanAccessPeer(&can, dev, attr.device));
                if (!can) throw Unsupported("device group: the block lives on GPU " + std::to_string(attr.device) + ", which GPU " +
                                            std::to_string(dev) + " cannot read (no peer access): upload it from host memory");
                HIP_CHECK(hipSetDevice(dev));
                const hipError_t e = hipDeviceEnablePeerAccess(attr.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
                (void)hipGetLastError();
            }
        }
        each_shard(ctx, [&](int s, mlhip_ctx* c) {
            const uint64_t lo = gd->first_row[(size_t)s];
            gd->parts[(size_t)s] = upload_common(c, x ? x + (int64_t)lo * ld : nullptr, on_device, d, rows_of(gd, s), ld);
        });
        gd->shift = gd->parts[0]->shift;
        require(gd->parts[0]->n_global == n, "device group: the shards disagree on the sample size");
    } catch (...) {
        delete gd;
        throw;
    }
    ctx->adopt(gd);
    return gd;
}

void sample_covariance(mlhip_ctx* ctx, mlhip_data* data, double* mean, double* covariance)
{
    check_group_data(ctx, data);
    require(covariance, "null argument");
    const size_t d = (size_t)data->d;
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        std::vector<double> m(d), cov(d * d);                 // every shard holds the same all-reduced result: shard 0's goes out
        check_status(mlhip_sample_covariance(c, data->parts[(size_t)s], s == 0 && mean ? mean : m.data(), s == 0 ? covariance : cov.data()));
    });
}

void xxt_xy(mlhip_ctx* ctx, mlhip_data* data, const double* y, double* xxt, double* xy)
{
    check_group_data(ctx, data);
    require((y || data->n_global == 0) && xxt && xy, "null argument");
    const size_t d = (size_t)data->d;
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        std::vector<double> a(s ? d * d : 0), b(s ? d : 0);
        check_status(mlhip_xxt_xy(c, data->parts[(size_t)s], y ? y + data->first_row[(size_t)s] : nullptr, s ? a.data() : xxt, s ? b.data() : xy));
    });
}

void random_partition_means(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const uint32_t* order, const uint32_t* offsets, double* means,
                            double* sizes)
{
    check_group_data(ctx, data);
    mlhip_group* g = group_of(ctx);
    require(K >= 1 && offsets && means && sizes && (order || data->n_global == 0), "null argument");
    require(offsets[0] == 0 && offsets[K] == data->n_global, "offsets must cover the sample's rows");
    for (uint32_t k = 0; k < K; ++k) require(offsets[k] <= offsets[k + 1], "offsets must ascend");
    // The running means are order dependent (ML/Clustering.cpp:33-35): the shards continue each other's K d chains in row order,
    // one after the other. A cluster's rows are listed ascending, so a shard's share of every list is one contiguous piece.
    std::vector<uint32_t> local_order, local_offsets((size_t)K + 1);
    for (int s = 0; s < g->n; ++s) {
        const uint64_t lo = data->first_row[(size_t)s], hi = data->first_row[(size_t)s + 1];
        local_order.clear();
        for (uint32_t k = 0; k < K; ++k) {
            local_offsets[k] = (uint32_t)local_order.size();
            const uint32_t* b = order + offsets[k];
            const uint32_t* e = order + offsets[k + 1];
            const uint32_t* from = std::lower_bound(b, e, lo, [](uint32_t v, uint64_t bound) { return (uint64_t)v < bound; });
            const uint32_t* to = std::lower_bound(from, e, hi, [](uint32_t v, uint64_t bound) { return (uint64_t)v < bound; });
            for (const uint32_t* p = from; p < to; ++p) local_order.push_back((uint32_t)(*p - lo));
        }
        local_offsets[K] = (uint32_t)local_order.size();
        require(local_order.size() == hi - lo, "order must list every row exactly once, ascending within a cluster");
        check_status(mlhip_random_partition_means(g->shard[(size_t)s], data->parts[(size_t)s], K, local_order.data(), local_offsets.data(),
                                                  means, sizes));
    }
}

namespace {
/// Parameter sets of the shards: inputs may alias outputs and every shard writes its (identical) results, so each works on copies
/// of its own; shard 0's copy goes back to the caller.
struct ShardParams {
    std::vector<double> mixing, means, covs;
    double ll = 0;
};
}  // namespace

void em_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, bool diag, const double* mixing, const double* means, const double* covs,
             double* log_likelihood, double* mixing_out, double* means_out, double* covs_out)
{
    check_group_data(ctx, data);
    require(K >= 1, "At least one component required");
    require(mixing && means && covs && log_likelihood && mixing_out && means_out && covs_out, "null argument");
    mlhip_group* g = group_of(ctx);
    const size_t d = (size_t)data->d, n_cov = diag ? K * d : K * d * d;
    std::vector<ShardParams> p((size_t)g->n);
    for (auto& q : p) { q.mixing.assign(mixing, mixing + K); q.means.assign(means, means + K * d); q.covs.assign(covs, covs + n_cov); }
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        ShardParams& q = p[(size_t)s];
        check_status(diag ? mlhip_em_step_diag(c, data->parts[(size_t)s], K, q.mixing.data(), q.means.data(), q.covs.data(), &q.ll,
                                               q.mixing.data(), q.means.data(), q.covs.data())
                          : mlhip_em_step(c, data->parts[(size_t)s], K, q.mixing.data(), q.means.data(), q.covs.data(), &q.ll,
                                          q.mixing.data(), q.means.data(), q.covs.data()));
    });
    std::copy(p[0].mixing.begin(), p[0].mixing.end(), mixing_out);
    std::copy(p[0].means.begin(), p[0].means.end(), means_out);
    std::copy(p[0].covs.begin(), p[0].covs.end(), covs_out);
    *log_likelihood = p[0].ll;
}

void em_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, int covariance_type, double* mixing, double* means, double* covs,
                uint32_t max_steps, double atol, double rtol, uint32_t* steps_done, int* converged, double* log_likelihood,
                double* history)
{
    check_group_data(ctx, data);
    require(K >= 1, "At least one component required");
    require(mixing && means && covs && steps_done && converged && log_likelihood, "null argument");
    require(covariance_type == MLHIP_COVARIANCE_FULL || covariance_type == MLHIP_COVARIANCE_DIAGONAL, "bad covariance_type");
    mlhip_group* g = group_of(ctx);
    const size_t d = (size_t)data->d, n_cov = covariance_type == MLHIP_COVARIANCE_DIAGONAL ? K * d : K * d * d;
    std::vector<ShardParams> p((size_t)g->n);
    for (size_t s = 1; s < p.size(); ++s) {
        p[s].mixing.assign(mixing, mixing + K); p[s].means.assign(means, means + K * d); p[s].covs.assign(covs, covs + n_cov);
    }
    std::vector<uint32_t> steps((size_t)g->n, 0);
    std::vector<int> conv((size_t)g->n, 0);
    // (the ranks' end-of-fit checksum exchange inside mlhip_em_iterate holds the shards to bit-identical parameters)
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        ShardParams& q = p[(size_t)s];
        std::vector<double> own_history(s && history ? max_steps : 0);
        check_status(mlhip_em_iterate(c, data->parts[(size_t)s], K, covariance_type, s ? q.mixing.data() : mixing, s ? q.means.data() : means,
                                      s ? q.covs.data() : covs, max_steps, atol, rtol, &steps[(size_t)s], &conv[(size_t)s], &q.ll,
                                      s ? (history ? own_history.data() : nullptr) : history));
    });
    for (int s = 1; s < g->n; ++s)
        if (steps[(size_t)s] != steps[0] || conv[(size_t)s] != conv[0]) throw std::runtime_error("device group: the shards stopped at different iterations");
    *steps_done = steps[0];
    *converged = conv[0];
    *log_likelihood = p[0].ll;
}

void em_expectation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* mixing, const double* means, const double* covs,
                    double* log_likelihood)
{
    check_group_data(ctx, data);
    require(mixing && means && covs && log_likelihood, "null argument");
    std::vector<double> ll((size_t)group_of(ctx)->n, 0.0);
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        check_status(mlhip_em_expectation(c, data->parts[(size_t)s], K, mixing, means, covs, &ll[(size_t)s]));
    });
    *log_likelihood = ll[0];
}

void em_maximisation(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, int source, const double* resp, int64_t ldr, const uint32_t* labels,
                     double* mixing_out, double* means_out, double* covs_out)
{
    check_group_data(ctx, data);
    require(K >= 1, "At least one component required");
    require(mixing_out && means_out && covs_out, "null argument");
    if (source == 1) {
        require(resp || data->n_global == 0, "null argument");
        require(ldr >= 0 && (uint64_t)ldr >= data->n_global, "ldr must be >= the number of rows");
    }
    if (source == 2) require(labels || data->n_global == 0, "null argument");
    mlhip_group* g = group_of(ctx);
    const size_t d = (size_t)data->d;
    std::vector<ShardParams> p((size_t)g->n);
    for (size_t s = 1; s < p.size(); ++s) { p[s].mixing.resize(K); p[s].means.resize(K * d); p[s].covs.resize(K * d * d); }
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        ShardParams& q = p[(size_t)s];
        double* pi = s ? q.mixing.data() : mixing_out;
        double* mu = s ? q.means.data() : means_out;
        double* cv = s ? q.covs.data() : covs_out;
        mlhip_data* part = data->parts[(size_t)s];
        const uint64_t lo = data->first_row[(size_t)s];
        if (source == 0) check_status(mlhip_em_maximisation(c, part, K, pi, mu, cv));
        else if (source == 1) check_status(mlhip_em_maximisation_from(c, part, K, resp ? resp + lo : nullptr, ldr, pi, mu, cv));
        else check_status(mlhip_em_maximisation_from_labels(c, part, K, labels ? labels + lo : nullptr, pi, mu, cv));
    });
}

void em_responsibilities(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* resp, int64_t ldr, uint64_t first, uint64_t count)
{
    check_group_data(ctx, data);
    require(first <= data->n_global && count <= data->n_global - first, "row range beyond the sample");
    require(resp || count == 0, "null argument");
    require(ldr >= 0 && (uint64_t)ldr >= count, "ldr must be >= the number of rows");
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        const uint64_t lo = std::max(first, data->first_row[(size_t)s]), hi = std::min(first + count, data->first_row[(size_t)s + 1]);
        if (lo >= hi) return;                                 // (no collective inside: a shard may sit this one out)
        check_status(mlhip_em_responsibilities_rows(c, data->parts[(size_t)s], K, lo - data->first_row[(size_t)s], hi - lo, resp + (lo - first), ldr));
    });
}

void em_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, uint32_t* labels)
{
    check_group_data(ctx, data);
    require(labels || data->n_global == 0, "null argument");
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        check_status(mlhip_em_labels(c, data->parts[(size_t)s], K, labels ? labels + data->first_row[(size_t)s] : nullptr));
    });
}

void kmeans_step(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, bool accumulate, const double* centroids, double* inertia,
                 uint64_t* n_changed, double* counts, double* centroids_out)
{
    check_group_data(ctx, data);
    require(K >= 1, "At least one component required");
    require(centroids && inertia && n_changed && (!accumulate || (counts && centroids_out)), "null argument");
    mlhip_group* g = group_of(ctx);
    const size_t kd = (size_t)K * data->d;
    const std::vector<double> in(centroids, centroids + kd);   // (centroids_out may alias centroids)
    std::vector<double> inert((size_t)g->n, 0.0);
    std::vector<uint64_t> changed((size_t)g->n, 0);
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        mlhip_data* part = data->parts[(size_t)s];
        if (accumulate) {
            std::vector<double> cnt(s ? K : 0), out(s ? kd : 0);
            check_status(mlhip_kmeans_step(c, part, K, in.data(), &inert[(size_t)s], &changed[(size_t)s], s ? cnt.data() : counts,
                                           s ? out.data() : centroids_out));
        } else {
            check_status(mlhip_kmeans_assign(c, part, K, in.data(), &inert[(size_t)s], &changed[(size_t)s]));
        }
    });
    *inertia = inert[0];
    *n_changed = changed[0];
}

void kmeans_iterate(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, double* centroids, double* old_centroids, uint32_t max_steps,
                    double atol, uint32_t* steps_done, int* converged, double* inertia, double* counts)
{
    check_group_data(ctx, data);
    require(K >= 1, "At least one component required");
    require(centroids && steps_done && converged && inertia, "null argument");
    mlhip_group* g = group_of(ctx);
    const size_t kd = (size_t)K * data->d;
    const std::vector<double> start(centroids, centroids + kd);
    std::vector<uint32_t> steps((size_t)g->n, 0);
    std::vector<int> conv((size_t)g->n, 0);
    std::vector<double> inert((size_t)g->n, 0.0);
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        std::vector<double> cur(s ? start : std::vector<double>()), old(s ? kd : 0), cnt(s ? K : 0);
        check_status(mlhip_kmeans_iterate(c, data->parts[(size_t)s], K, s ? cur.data() : centroids, s ? old.data() : old_centroids, max_steps,
                                          atol, &steps[(size_t)s], &conv[(size_t)s], &inert[(size_t)s], s ? cnt.data() : counts));
    });
    for (int s = 1; s < g->n; ++s)
        if (steps[(size_t)s] != steps[0] || conv[(size_t)s] != conv[0]) throw std::runtime_error("device group: the shards stopped at different steps");
    *steps_done = steps[0];
    *converged = conv[0];
    *inertia = inert[0];
}

void kmeans_labels(mlhip_ctx* ctx, mlhip_data* data, uint32_t* labels)
{
    check_group_data(ctx, data);
    require(labels || data->n_global == 0, "null argument");
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        check_status(mlhip_kmeans_labels(c, data->parts[(size_t)s], labels ? labels + data->first_row[(size_t)s] : nullptr));
    });
}

void kmeans_distances(mlhip_ctx* ctx, mlhip_data* data, double* dist2)
{
    check_group_data(ctx, data);
    require(dist2 || data->n_global == 0, "null argument");
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        check_status(mlhip_kmeans_distances(c, data->parts[(size_t)s], dist2 ? dist2 + data->first_row[(size_t)s] : nullptr));
    });
}

void kpp_draw(mlhip_ctx* ctx, mlhip_data* data, const double* centroid, int first, double u, uint64_t first_row, uint64_t* index,
              int* certain, double* weights_out)
{
    check_group_data(ctx, data);
    require(centroid && index && certain, "null argument");
    require(first_row == 0, "a device group holds the whole sample: first_row must be 0");
    mlhip_group* g = group_of(ctx);
    std::vector<uint64_t> idx((size_t)g->n, 0);
    std::vector<int> sure((size_t)g->n, 0);
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        const uint64_t lo = data->first_row[(size_t)s];
        check_status(mlhip_kpp_draw(c, data->parts[(size_t)s], centroid, first, u, lo, &idx[(size_t)s], &sure[(size_t)s],
                                    weights_out ? weights_out + lo : nullptr));
    });
    *index = idx[0];
    *certain = sure[0];
}

void kpp_weights(mlhip_ctx* ctx, mlhip_data* data, double* weights_out)
{
    check_group_data(ctx, data);
    require(weights_out || data->n_global == 0, "null argument");
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        check_status(mlhip_kpp_weights(c, data->parts[(size_t)s], weights_out ? weights_out + data->first_row[(size_t)s] : nullptr));
    });
}

void min_squared_distances(mlhip_ctx* ctx, mlhip_data* data, uint32_t K, const double* centroids, double* dist2)
{
    check_group_data(ctx, data);
    require(centroids && (dist2 || data->n_global == 0), "null argument");
    each_shard(ctx, [&](int s, mlhip_ctx* c) {
        check_status(mlhip_min_squared_distances(c, data->parts[(size_t)s], K, centroids, dist2 ? dist2 + data->first_row[(size_t)s] : nullptr));
    });
}

void timing_enable(mlhip_ctx* ctx, int on)
{
    for (mlhip_ctx* c : group_of(ctx)->shard) check_status(mlhip_timing_enable(c, on));
}

void timing_reset(mlhip_ctx* ctx)
{
    for (mlhip_ctx* c : group_of(ctx)->shard) check_status(mlhip_timing_reset(c));
}

/// The slowest shard's average (launch counts of that shard): what bounds the group's step.
void timing_get(mlhip_ctx* ctx, const char* name, double* avg_ms, uint64_t* launches)
{
    *avg_ms = 0;
    *launches = 0;
    for (mlhip_ctx* c : group_of(ctx)->shard) {
        double ms = 0;
        uint64_t cnt = 0;
        check_status(mlhip_timing_get(c, name, &ms, &cnt));
        if (cnt && ms >= *avg_ms) { *avg_ms = ms; *launches = cnt; }
    }
}

}  // namespace grp
}  // namespace mlhip_rt
