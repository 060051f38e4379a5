#pragma once
/* Which GPU the facade classes run on. Not part of the reference API (the reference is CPU-only); device selection
 * is deliberately kept out of the model classes so that ml::EM / ml::Clustering::KMeans keep their signatures. */
#include "dll.hpp"

struct mlhip_ctx;

namespace ml {
namespace device {
/** Process-wide context used by the facade, created on first use: a device GROUP over several GPUs when the environment asks
for one (MLHIP_DEVICES=0,1,2,3 or MLHIP_NUM_GPUS=8: `fit` then row-shards its one data block over them, mlhip_ctx_create_group),
else one GPU (MLHIP_DEVICE, else LOCAL_RANK, else 0).
@throw std::runtime_error If no HIP device is usable -- there is no CPU fallback. */
DLL_DECLSPEC mlhip_ctx* context();
/** The context the facade would use right now, WITHOUT creating one: the override, else the default if it already
exists, else nullptr. Lets `fit` see a multi-rank (row-sharded) context before deciding anything from the local shard size. */
DLL_DECLSPEC mlhip_ctx* peek_context();
/** Replaces the process-wide context (not owned); nullptr restores the lazily created default. */
DLL_DECLSPEC void set_context(mlhip_ctx* ctx);
/** Throws the C++ exception matching a failed C-ABI call (std::invalid_argument / std::domain_error / std::runtime_error). */
DLL_DECLSPEC void check(int mlhip_status);
}
}
