// Normalised responsibilities and hard labels from the log-responsibilities an E-step left in HBM --
// replaces the row normalisation of EM::expectation_step (reference ML/EM.cpp:214-218) and
// EM::calculate_labels (ML/EM.cpp:289-304: strict '>', components scanned in ascending order, so the
// first maximum wins). Labels are taken from the very r values that are stored, like the reference does.
#include "device.hpp"
#include "exp_nonpos.hpp"

namespace mlhip {
namespace {

__global__ __launch_bounds__(256) void em_resp_kernel(const double* __restrict__ lw, size_t ldr,
                                                       const double* __restrict__ lse, uint32_t n, int K,
                                                       double* __restrict__ resp, size_t ldo,
                                                       uint32_t* __restrict__ labels)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const double l = lse[i];
        double best = -1.0;
        uint32_t arg = 0xffffffffu;   // ML/EM.cpp:294 starts from label -1
        for (int k = 0; k < K; ++k) {
            const double r = exp_nonpos(lw[(size_t)k * ldr + i] - l);
            if (resp) resp[(size_t)k * ldo + i] = r;
            if (r > best) { best = r; arg = (uint32_t)k; }
        }
        if (labels) labels[i] = arg;
    }
}

}  // namespace

void launch_em_responsibilities(const RespArgs& a, hipStream_t stream)
{
    uint32_t blocks = (a.n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(em_resp_kernel, dim3(blocks), dim3(256), 0, stream, a.lw, a.ldr, a.lse, a.n, a.K, a.resp, a.ldo,
                       a.labels);
}

}  // namespace mlhip
