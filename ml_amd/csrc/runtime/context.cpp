// Context family of the C ABI (include/mlhip.h): device / stream, the statistics all-reduce (caller hook or the library's own
// RCCL communicator, dlopen'ed), error slot, kernel timing.
#include "internal.hpp"

namespace mlhip_rt {
thread_local std::string g_error;
}

namespace mlhip_rt {


/// A device group is ONE rank to its caller (its shards meet inside the library): it takes no hook or communicator of its own.
static void refuse_group(const mlhip_ctx* ctx)
{
    if (ctx->group || ctx->member_of)
        throw Unsupported("a device group sums its shards' statistics itself: it cannot be given an all-reduce hook or a communicator");
}


int env_int(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
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


mlhip_ctx* create_single_context(int device_id)
{
    auto* ctx = new mlhip_ctx;
    try {
        ctx->device = device_id;
        ctx->use();
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        // The first copy between PINNED host memory and the device sets up the runtime's DMA path (8.4 ms on the MI355X box, rocprofv3
        // --hip-trace of tools/first_call.py: inside the first fit's upload; a copy from pageable memory does not take that path).
        // Every fit uploads its samples through the pinned staging pair and reads results back the same way, so both directions
        // are touched here, once per context, next to the ~150 ms the device initialisation takes anyway; the two small blocks
        // stay with the context as the first staging pair.
        {
            const size_t probe = 64 * 1024;
            ctx->up_pin[0].reserve(probe);
            ctx->up_stage[0].reserve(probe);
            std::memset(ctx->up_pin[0].p, 0, probe);
            HIP_CHECK(hipMemcpyAsync(ctx->up_stage[0].p, ctx->up_pin[0].p, probe, hipMemcpyHostToDevice, ctx->stream));
            HIP_CHECK(hipMemcpyAsync(ctx->up_pin[0].p, ctx->up_stage[0].p, probe, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
    } catch (...) {
        delete ctx;
        throw;
    }
    return ctx;
}


void destroy_single_context(mlhip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) drop_rccl(ctx);
    {
        std::lock_guard<std::mutex> lock(ctx->handles_m);
        for (mlhip_data* h : ctx->handles) {          // (handles that outlive their context: see mlhip_ctx::handles)
            h->attach_pool(nullptr);
            h->ctx = nullptr;
        }
        ctx->handles.clear();
    }
    ctx->small_dev.release();
    ctx->small_host.release();
    for (int b = 0; b < 2; ++b) { ctx->up_stage[b].release(); ctx->up_pin[b].release(); }
    ctx->pool.drain();
    for (auto& p : ctx->pending) ctx->spare_events.push_back(p.second);
    for (auto& e : ctx->spare_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

}  // namespace mlhip_rt

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
        *out = create_single_context(device_id);
    });
}

int mlhip_ctx_create_group(int n_shards, const int* device_ids, mlhip_ctx** out)
{
    return guarded([&] {
        require(out != nullptr, "null out");
        *out = grp::create(n_shards, device_ids);
    });
}

int mlhip_ctx_create_default(mlhip_ctx** out)
{
    return guarded([&] {
        require(out != nullptr, "null out");
        // MLHIP_DEVICES=0,1,2,3 (one shard per entry; an entry may repeat) or MLHIP_NUM_GPUS=n (shard s on GPU s mod the number of
        // GPUs visible) ask for a device group; a process started by a one-process-per-GPU launcher (LOCAL_RANK set) keeps its one
        // GPU unless MLHIP_DEVICES names a list.
        std::vector<int> ids;
        if (const char* e = std::getenv("MLHIP_DEVICES")) {
            for (const char* p = e; *p;) {
                char* end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) throw InvalidArgument("MLHIP_DEVICES must be a comma-separated list of GPU indices");
                ids.push_back((int)v);
                p = end;
                while (*p == ',' || *p == ' ') ++p;
            }
        }
        int shards = (int)ids.size();
        if (!shards && !std::getenv("LOCAL_RANK")) shards = env_int("MLHIP_NUM_GPUS", 0);
        if (shards >= 2 || ids.size() == 1) {
            *out = grp::create(shards, ids.empty() ? nullptr : ids.data());
            return;
        }
        check_status(mlhip_ctx_create(-1, out));
    });
}

int mlhip_ctx_destroy(mlhip_ctx* ctx)
{
    return guarded([&] {
        if (!ctx) return;
        if (ctx->group) { grp::destroy(ctx); return; }
        require(!ctx->member_of, "a shard context belongs to its device group");
        destroy_single_context(ctx);
    });
}

int mlhip_ctx_synchronize(mlhip_ctx* ctx)
{
    return guarded([&] {
        require(ctx, "null context");
        if (ctx->group) { grp::synchronize(ctx); return; }
        ctx->use();
        ctx->sync();
    });
}
int mlhip_ctx_device(const mlhip_ctx* ctx, int* device_id)
{
    return guarded([&] { require(ctx && device_id, "null argument"); *device_id = ctx->device; });
}
int mlhip_ctx_stream(const mlhip_ctx* ctx, void** stream)
{
    return guarded([&] {
        require(ctx && stream, "null argument");
        *stream = (void*)(ctx->group ? grp::shard_context(ctx, 0)->stream : ctx->stream);   // (a group: its first shard's)
    });
}

int mlhip_ctx_shards(const mlhip_ctx* ctx, int* n_shards)
{
    return guarded([&] { require(ctx && n_shards, "null argument"); *n_shards = grp::shard_count(ctx); });
}
int mlhip_ctx_shard_device(const mlhip_ctx* ctx, int shard, int* device_id)
{
    return guarded([&] {
        require(ctx && device_id, "null argument");
        if (ctx->group) { *device_id = grp::shard_context(ctx, shard)->device; return; }
        require(shard == 0, "no such shard");
        *device_id = ctx->device;
    });
}
int mlhip_ctx_reduce_kind(const mlhip_ctx* ctx, const char** kind)
{
    return guarded([&] { require(ctx && kind, "null argument"); *kind = grp::reduce_kind(ctx); });
}

int mlhip_ctx_set_allreduce(mlhip_ctx* ctx, mlhip_allreduce_fn fn, void* user, int on_device, int world_size, int rank)
{
    return guarded([&] {
        require(ctx, "null context");
        refuse_group(ctx);
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
        refuse_group(ctx);
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        init_rccl(ctx, id, world_size, rank);
    });
}

int mlhip_ctx_init_rccl_file(mlhip_ctx* ctx, const char* path, int world_size, int rank)
{
    return guarded([&] {
        require(ctx && path && *path, "null argument");
        refuse_group(ctx);
        require(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size / rank");
        // The file carries [job nonce | unique id]. The nonce is a hash of what tells this job's launch from another's
        // (MLHIP_RCCL_NONCE, else the launcher's TORCHELASTIC_RUN_ID / MASTER_ADDR / MASTER_PORT): a waiting rank rejects the
        // left-over file of a job that died before its rank 0 removed it, even inside the staleness window.
        uint64_t nonce = 1469598103934665603ull;
        for (const char* name : {"MLHIP_RCCL_NONCE", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"}) {
            const char* v = std::getenv(name);
            for (const char* c = v ? v : ""; *c; ++c) { nonce ^= (unsigned char)*c; nonce *= 1099511628211ull; }
            nonce ^= 0xff; nonce *= 1099511628211ull;
        }
        struct Record { uint64_t nonce; ncclUniqueId id; } rec;
        ncclUniqueId id;
        if (rank == 0) {
            const Rccl& r = Rccl::get();
            r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
            rec.nonce = nonce;
            rec.id = id;
            std::remove(path);                                      // (an earlier job's left-over)
            const std::string tmp = std::string(path) + ".tmp";     // written whole, then renamed: readers never see a part
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&rec, 1, sizeof rec, f) != sizeof rec || std::fclose(f) != 0 || std::rename(tmp.c_str(), path) != 0)
                throw std::runtime_error(std::string("cannot write the RCCL rendezvous file ") + path);
        } else {
            const int limit_s = std::max(1, env_int("MLHIP_RCCL_TIMEOUT_S", 120));
            const int stale_s = std::max(1, env_int("MLHIP_RCCL_STALE_S", 600));
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                // neither the left-over of an old job (age) nor the file of another launch (nonce) is taken for this job's
                struct stat st;
                const bool fresh = ::stat(path, &st) == 0 && std::time(nullptr) - st.st_mtime <= stale_s;
                if (FILE* f = fresh ? std::fopen(path, "rb") : nullptr) {
                    const size_t got = std::fread(&rec, 1, sizeof rec, f);
                    std::fclose(f);
                    if (got == sizeof rec && rec.nonce == nonce) { id = rec.id; break; }
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
        if (ctx->group) ctx = grp::shard_context(ctx, 0);       // (a group over distinct GPUs: the communicator of ncclCommInitAll)
        if (!ctx->comm) return;
        const Rccl& r = Rccl::get();
        r.check(r.CommCount(ctx->comm, nranks), "ncclCommCount");
    });
}

int mlhip_ctx_finalize_rccl(mlhip_ctx* ctx)
{
    return guarded([&] { require(ctx, "null context"); refuse_group(ctx); ctx->use(); drop_rccl(ctx); });
}

int mlhip_ctx_allreduce(mlhip_ctx* ctx, double* buf, size_t count)
{
    return guarded([&] {
        require(ctx && (buf || count == 0), "null argument");
        if (ctx->group) return;                 // (a group is one rank to its caller: nothing to sum with)
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

int mlhip_timing_enable(mlhip_ctx* ctx, int on)
{
    return guarded([&] {
        require(ctx, "null context");
        if (ctx->group) { grp::timing_enable(ctx, on); return; }
        ctx->use();
        if (!on) ctx->resolve_timers();
        ctx->timing = on != 0;
    });
}
int mlhip_timing_reset(mlhip_ctx* ctx)
{
    return guarded([&] {
        require(ctx, "null context");
        if (ctx->group) { grp::timing_reset(ctx); return; }
        ctx->use();
        ctx->resolve_timers();
        ctx->timers.clear();
    });
}
int mlhip_timing_get(mlhip_ctx* ctx, const char* name, double* avg_ms, uint64_t* launches)
{
    return guarded([&] {
        require(ctx && name && avg_ms && launches, "null argument");
        if (ctx->group) { grp::timing_get(ctx, name, avg_ms, launches); return; }
        ctx->use();
        ctx->resolve_timers();
        auto it = ctx->timers.find(name);
        if (it == ctx->timers.end() || it->second.launches == 0) { *avg_ms = 0; *launches = 0; return; }
        *avg_ms = it->second.total_ms / (double)it->second.launches;
        *launches = it->second.launches;
    });
}

}  // extern "C"
