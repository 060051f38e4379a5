// Empirical lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D[4x4] = A[4x4] B[4x4] + C): for every pair of lanes
// (la, lb) run the instruction with A = e_la, B = e_lb and record which lane of D becomes 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* where)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) where[la * 64 + lb] = lane;
        }
}
int main()
{
    int* dev;
    hipMalloc(&dev, 4096 * sizeof(int));
    hipMemset(dev, 0xff, 4096 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dev);
    std::vector<int> w(4096);
    hipMemcpy(w.data(), dev, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // For each A lane: the set of B lanes it pairs with and the D lanes hit.
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (w[la * 64 + lb] >= 0) printf(" (B%d->D%d)", lb, w[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
