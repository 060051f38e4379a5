// Sustained fp64 matrix-core rate with REAL operands: tools/microbench_fp64 measures ~1 ms bursts on near-constant operands
// (76 TFLOP/s for v_mfma_f64_4x4x4_4b), but the EM kernels run tens of milliseconds back to back on random data, where the
// chip's power management sets the clock (MI355X_MICROARCH.md: zero / constant operands run up to +19 % faster than random
// ones at the same cycle count). This tool keeps every SIMD issuing dependent-free MFMAs on N(0,1) operands for a few seconds
// and reports the rate of the second half -- the ceiling the E-step / statistics kernels are priced against in DESIGN.md.
// Build: hipcc -O3 --offload-arch=gfx950 microbench_sustained.hip -o microbench_sustained
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int ITERS = 2048;

template <bool SMALL>
__global__ __launch_bounds__(256) void mfma_rand(const double* __restrict__ ops, double* out)
{
    double av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        av[i] = ops[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
        bv[i] = ops[(blockIdx.x * 8 + 4 + i) * 256 + threadIdx.x];
    }
    double s = 0;
    if constexpr (SMALL) {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i & 3], bv[i >> 2], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        d4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
        for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i & 3], bv[i >> 2], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool SMALL> int run(const char* name, const double* ops, double* out, int grid, double seconds)
{
    const double flop_per_launch = SMALL ? (double)grid * 4 * ITERS * 16 * 512.0 : (double)grid * 4 * (ITERS / 4) * 16 * 2048.0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    int phase = 0;
    double rate[2] = {0, 0};
    for (phase = 0; phase < 2; ++phase) {
        int launches = 0;
        CHECK(hipEventRecord(e0));
        const auto p0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count() < seconds / 2) {
            for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(mfma_rand<SMALL>, dim3(grid), dim3(256), 0, 0, ops, out);
            launches += 20;
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        rate[phase] = flop_per_launch * launches / (ms * 1e-3) / 1e12;
    }
    (void)t0;
    printf("%-22s random N(0,1) operands, 2 waves/SIMD, %4.1f s: first half %6.2f TFLOP/s, second half (sustained) %6.2f TFLOP/s\n", name,
           seconds, rate[0], rate[1]);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 2;          // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    std::vector<double> h((size_t)grid * 8 * 256);
    std::mt19937_64 gen(7);
    std::normal_distribution<double> nrm;
    for (double& v : h) v = nrm(gen);
    double *ops, *out;
    CHECK(hipMalloc(&ops, h.size() * sizeof(double)));
    CHECK(hipMalloc(&out, (size_t)grid * 256 * sizeof(double)));
    CHECK(hipMemcpy(ops, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    printf("device %s, %d CUs, clock %d MHz (spec fp64 matrix peak 78.6 TFLOP/s at 2400 MHz)\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    if (run<true>("v_mfma_f64_4x4x4_4b", ops, out, grid, 4.0)) return 1;
    if (run<false>("v_mfma_f64_16x16x4", ops, out, grid, 4.0)) return 1;
    return 0;
}
