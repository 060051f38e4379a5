// Microbenchmarks that pin the fp64 roofline numbers used in DESIGN.md / bench.py on the box at hand:
// v_fma_f64 rate, v_mfma_f64_16x16x4_f64 rate, both pipes together, and HBM streaming read rate.
// Build: hipcc -O3 --offload-arch=gfx950 microbench_fp64.hip -o microbench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ __launch_bounds__(256) void fma_kernel(double* out, double a, double b)
{
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_kernel(double* out, double a, double b)
{
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma4_kernel(double* out, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0-3: MFMA, waves 4-7: FMA (512-thread workgroup, 2 waves per SIMD)
__global__ __launch_bounds__(512) void both_kernel(double* out, double a, double b)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double s = 0;
    if (wave < 4) {
        d4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
        double av = a + threadIdx.x * 1e-9, bv = b;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-9 + i;
        for (int it = 0; it < ITERS * 8; ++it) {   // 8 MFMA (64 cyc each?) vs 16 FMA (4 cyc each): scale to similar duration
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void read_kernel(const double2* __restrict__ in, size_t n2, double* out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 v = in[i];
        s += v.x + v.y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F> float time_ms(F&& f, int reps = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    double* out;
    CHECK(hipMalloc(&out, sizeof(double) * 512 * cus * 16));

    for (int wps : {1, 2, 4}) {   // waves per SIMD
        const int blocks = cus * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); });
        const double flop = 2.0 * 16 * ITERS * 256.0 * blocks;
        printf("v_fma_f64            %d waves/SIMD: %7.3f ms  %7.2f TFLOP/s\n", wps, ms, flop / ms * 1e-9);
    }
    for (int wps : {1, 2}) {
        const int blocks = cus * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); });
        const double flop = 2.0 * 16 * 16 * 4 * 8 * ITERS * 4.0 * blocks;
        printf("v_mfma_f64_16x16x4   %d waves/SIMD (8 acc): %7.3f ms  %7.2f TFLOP/s\n", wps, ms, flop / ms * 1e-9);
    }
    {
        const int blocks = cus;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); });
        const double flop = 2.0 * 16 * 16 * 4 * 1 * ITERS * 4.0 * blocks;
        printf("v_mfma_f64_16x16x4   1 wave/SIMD (1 acc, dependent chain): %7.3f ms  %7.2f TFLOP/s\n", ms, flop / ms * 1e-9);
    }
    for (int wps : {1, 2}) {
        const int blocks = cus * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma4_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); });
        const double flop = 2.0 * 4 * 4 * 4 * 4 * 8 * ITERS * 4.0 * blocks;   // 4 blocks of 4x4x4 per instruction
        printf("v_mfma_f64_4x4x4_4b  %d waves/SIMD (8 acc): %7.3f ms  %7.2f TFLOP/s\n", wps, ms, flop / ms * 1e-9);
    }
    {
        const int blocks = cus;
        float ms = time_ms([&] { hipLaunchKernelGGL(both_kernel, dim3(blocks), dim3(512), 0, 0, out, 1.0000001, 1e-9); });
        const double flop_m = 2.0 * 16 * 16 * 4 * 8 * ITERS * 4.0 * blocks;
        const double flop_v = 2.0 * 16 * ITERS * 8 * 256.0 * blocks;
        printf("MFMA waves + FMA waves on the same SIMDs: %7.3f ms  mfma %7.2f + valu %7.2f = %7.2f TFLOP/s\n", ms,
               flop_m / ms * 1e-9, flop_v / ms * 1e-9, (flop_m + flop_v) / ms * 1e-9);
    }
    {
        const size_t bytes = (size_t)4 << 30;
        double2* buf;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        const int blocks = cus * 8;
        float ms = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, out); });
        printf("HBM streaming read (16 B/lane): %7.3f ms  %7.2f TB/s\n", ms, bytes / ms * 1e-9);
        hipFree(buf);
    }
    hipFree(out);
    return 0;
}
