// Issue model of a gfx950 SIMD running fp64 matrix instructions next to other instructions: what does ONE extra instruction
// of a given kind cost a stream of v_mfma_f64_4x4x4_4b_f64 (16 cycles each), with one or two such waves per SIMD, spread
// evenly between the matrix instructions or clustered behind them? (The E-step variants of DESIGN.md 3.3 differ only in
// the number and kind of the instructions that feed the matrix pipe; this prices them.)
//
// Build: hipcc -O3 --offload-arch=gfx950 microbench_issue.hip -o microbench_issue ;  run: ./microbench_issue
// Output per configuration: shader cycles per matrix instruction as seen by the pipe (wave duration / matrix instructions
// issued on the SIMD in that time); 16.0 is the pipe's own rate.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
constexpr int ITERS = 4096;
constexpr int NM = 16;   // matrix instructions per loop body

static double base_big[2] = {64.0, 64.0};   // cycles per bare 16x16x4 instruction as measured (one / two waves per SIMD)

enum Kind { NONE, MOV32, DPP32, FMA64, DSREAD, DPP64, SWAP16, SNOP, ADD64, KINDS };
static const char* kind_name[KINDS] = {"none", "v_mov_b32", "v_mov_b32_dpp", "v_fma_f64", "ds_read_b64", "v_mov_b64_dpp", "v_permlane16_swap", "s_nop 0", "v_add_f64"};

template <int KIND> __device__ __forceinline__ void extra(double& f, int& i0, int& i1, double& l, const double* lds)
{
    if constexpr (KIND == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(i0) : "v"(i1));
    if constexpr (KIND == DPP32) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(i0) : "v"(i1));
    if constexpr (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(f) : "v"(l));
    if constexpr (KIND == ADD64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(f) : "v"(l));
    if constexpr (KIND == DSREAD) asm volatile("ds_read_b64 %0, %1" : "=v"(l) : "v"(i1));
    if constexpr (KIND == DPP64) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(f) : "v"(l));
    if constexpr (KIND == SWAP16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(i0), "+v"(i1));
    if constexpr (KIND == SNOP) asm volatile("s_nop 0");
}

/// PER extra instructions of KIND per NM matrix instructions; SPREAD: one after every (NM / PER)-th, else all behind the last.
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND, int PER, bool SPREAD, bool BIG = false>
__global__ __launch_bounds__(256, 2) void issue_kernel(const double* __restrict__ ops, double* out, long long* ticks)
{
    __shared__ double lds[512];
    lds[threadIdx.x] = ops[threadIdx.x];
    lds[threadIdx.x + 256] = ops[threadIdx.x + 256];
    __syncthreads();
    double av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        av[i] = ops[(i * 256 + threadIdx.x) & 4095];
        bv[i] = ops[((4 + i) * 256 + threadIdx.x) & 4095];
    }
    double acc[NM];
    d4 big[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) { acc[i] = 0.0; big[i] = d4{0.0, 0.0, 0.0, 0.0}; }
    double f = 1.0, l = av[0];
    int i0 = threadIdx.x, i1 = (threadIdx.x & 63) * 8;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if constexpr (BIG) big[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i & 3], bv[i >> 2], big[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i & 3], bv[i >> 2], acc[i], 0, 0, 0);
            if constexpr (KIND != NONE && SPREAD) {
                constexpr int every = PER >= NM ? 1 : NM / PER;
                if (i % every == every - 1) {
#pragma unroll
                    for (int r = 0; r < (PER >= NM ? PER / NM : 1); ++r) extra<KIND>(f, i0, i1, l, lds);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KIND != NONE && !SPREAD) {
#pragma unroll
            for (int r = 0; r < PER; ++r) extra<KIND>(f, i0, i1, l, lds);
        }
        if constexpr (KIND == DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = f + l + i0 + i1;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i] + big[i][0] + big[i][1] + big[i][2] + big[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int PER, bool SPREAD, bool BIG = false>
int run(const double* ops, double* out, long long* ticks, int num_cus)
{
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = num_cus * wps;
        hipLaunchKernelGGL((issue_kernel<KIND, PER, SPREAD, BIG>), dim3(grid), dim3(256), 0, 0, ops, out, ticks);   // warm-up
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((issue_kernel<KIND, PER, SPREAD, BIG>), dim3(grid), dim3(256), 0, 0, ops, out, ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> t(grid * 4);
        CHECK(hipMemcpy(t.data(), ticks, sizeof(long long) * t.size(), hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end());
        const double med = (double)t[t.size() / 2];
        const double per_mfma = med / ((double)ITERS * NM * wps);
        const double tflops = (double)grid * 4 * ITERS * NM * (BIG ? 2048.0 : 512.0) * 20 / (ms * 1e-3) * 1e-12;
        printf("%s %-18s per16=%2d %-9s waves/SIMD=%d  cycles/mfma(pipe)=%6.2f  extra cycles per added instr=%6.2f  %6.1f TFLOP/s\n",
               BIG ? "16x16x4" : "4x4x4  ", kind_name[KIND], KIND == NONE ? 0 : PER, SPREAD ? "spread" : "clustered", wps, per_mfma,
               KIND == NONE ? 0.0 : (per_mfma - (BIG ? base_big[wps - 1] : 16.0)) * NM * wps / (PER * wps), tflops);
        if (BIG && KIND == NONE) base_big[wps - 1] = per_mfma;
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    return 0;
}

template <int KIND> int run_kind(const double* ops, double* out, long long* ticks, int num_cus)
{
    if (run<KIND, 4, true>(ops, out, ticks, num_cus)) return 1;
    if (run<KIND, 8, true>(ops, out, ticks, num_cus)) return 1;
    if (run<KIND, 16, true>(ops, out, ticks, num_cus)) return 1;
    if (run<KIND, 32, true>(ops, out, ticks, num_cus)) return 1;
    if (run<KIND, 8, false>(ops, out, ticks, num_cus)) return 1;
    if (run<KIND, 16, false>(ops, out, ticks, num_cus)) return 1;
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int num_cus = prop.multiProcessorCount;
    std::vector<double> h(4096);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(rng);
    double *ops, *out;
    long long* ticks;
    CHECK(hipMalloc(&ops, sizeof(double) * h.size()));
    CHECK(hipMalloc(&out, sizeof(double) * num_cus * 2 * 256));
    CHECK(hipMalloc(&ticks, sizeof(long long) * num_cus * 2 * 4));
    CHECK(hipMemcpy(ops, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    if (run<NONE, 4, true>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<MOV32>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<DPP32>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<DPP64>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<FMA64>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<ADD64>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<DSREAD>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<SWAP16>(ops, out, ticks, num_cus)) return 1;
    if (run_kind<SNOP>(ops, out, ticks, num_cus)) return 1;
    // the 16x16x4 shape (statistics kernel, K-means): one multiply per 4 matrix instructions is what em_mstats_wide interleaves
    if (run<NONE, 4, true, true>(ops, out, ticks, num_cus)) return 1;
    if (run<FMA64, 4, true, true>(ops, out, ticks, num_cus)) return 1;
    if (run<FMA64, 4, false, true>(ops, out, ticks, num_cus)) return 1;
    if (run<FMA64, 16, true, true>(ops, out, ticks, num_cus)) return 1;
    if (run<FMA64, 16, false, true>(ops, out, ticks, num_cus)) return 1;
    if (run<MOV32, 16, true, true>(ops, out, ticks, num_cus)) return 1;
    if (run<MOV32, 16, false, true>(ops, out, ticks, num_cus)) return 1;
    if (run<DSREAD, 16, true, true>(ops, out, ticks, num_cus)) return 1;
    return 0;
}
