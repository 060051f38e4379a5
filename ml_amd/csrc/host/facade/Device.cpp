#include "ML/Device.hpp"

#include <mutex>
#include <stdexcept>
#include <string>

#include "mlhip.h"

namespace ml {
namespace device {
namespace {
std::mutex g_mutex;
mlhip_ctx* g_default = nullptr;   // owned, created lazily, lives until process exit
mlhip_ctx* g_override = nullptr;  // not owned
}

void check(int status)
{
    if (status == MLHIP_OK) return;
    const std::string msg = mlhip_last_error();
    switch (status) {
    case MLHIP_E_INVALID_ARGUMENT: throw std::invalid_argument(msg);
    case MLHIP_E_DOMAIN: throw std::domain_error(msg);
    default: throw std::runtime_error(msg);
    }
}

mlhip_ctx* context()
{
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_override) return g_override;
    if (!g_default) check(mlhip_ctx_create_default(&g_default));   // (MLHIP_DEVICES / MLHIP_NUM_GPUS: a device group)
    return g_default;
}

mlhip_ctx* peek_context()
{
    std::lock_guard<std::mutex> lock(g_mutex);
    return g_override ? g_override : g_default;
}

void set_context(mlhip_ctx* ctx)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    g_override = ctx;
}
}  // namespace device
}  // namespace ml
