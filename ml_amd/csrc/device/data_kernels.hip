// Data-movement kernels: sample-major (the caller's d x N column-major block, reference
// ML/Clustering.hpp:28-33) -> dimension-major HBM layout, and deterministic column sums.
#include "device.hpp"

namespace mlhip {
namespace {

// Tile of 64 samples x d dims through LDS so that both the global read (consecutive doubles of a sample
// row) and the global write (consecutive samples of one dimension) are coalesced.
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, int64_t lds_, int d, uint64_t n,
                                                         double* __restrict__ dst, size_t ldd, uint64_t i0, int dc)
{
    extern __shared__ double tile[];   // [64][dc+1]: dimensions j0 .. j0 + dw of the tile's 64 samples (blockIdx.y: the chunk)
    const int j0 = blockIdx.y * dc, dw = min(dc, d - j0);
    const int DS = dc + 1;
    const uint64_t base = (uint64_t)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * dw; e += 256) {
        const int s = e / dw, j = e - s * dw;
        const uint64_t i = base + s;
        tile[s * DS + j] = i < n ? src[(int64_t)i * lds_ + j0 + j] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * dw; e += 256) {
        const int s = e & 63, j = e >> 6;
        const uint64_t i = base + s;
        if (i < n) dst[(size_t)(j0 + j) * ldd + i0 + i] = tile[s * DS + j];
    }
}

// Stage 1: block b of row j sums a contiguous chunk; stage 2: one block per row sums the 1024 partials.
__global__ __launch_bounds__(256) void colsum_stage1(const double* __restrict__ xt, size_t ldx, uint64_t n,
                                                      double* __restrict__ scratch)
{
    __shared__ double red[256];
    const int j = blockIdx.y;
    const uint64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    uint64_t hi = lo + chunk;
    if (hi > n) hi = n;
    double s = 0.0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) s += xt[(size_t)j * ldx + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) scratch[(size_t)j * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void colmax_stage1(const double* __restrict__ xt, size_t ldx, uint64_t n,
                                                      double* __restrict__ scratch)
{
    __shared__ double red[256];
    const int j = blockIdx.y;
    const uint64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    uint64_t hi = lo + chunk;
    if (hi > n) hi = n;
    double m = 0.0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) m = fmax(m, fabs(xt[(size_t)j * ldx + i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) scratch[(size_t)j * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void colmax_stage2(const double* __restrict__ scratch, int parts, double* __restrict__ out)
{
    __shared__ double red[256];
    const int j = blockIdx.x;
    double m = 0.0;
    for (int b = threadIdx.x; b < parts; b += 256) m = fmax(m, scratch[(size_t)j * parts + b]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = red[0];
}

__global__ __launch_bounds__(256) void colsum_stage2(const double* __restrict__ scratch, int parts, double* __restrict__ sums)
{
    __shared__ double red[256];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < parts; b += 256) s += scratch[(size_t)j * parts + b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[j] = red[0];
}

/// RandomPartition::init's running means (reference ML/Clustering.cpp:27-37): `c_k += (x_i - c_k) / ++n_k` over the rows of
/// cluster k in ROW ORDER. The K d chains are independent of each other; each is strictly sequential, and IEEE subtraction,
/// division and addition are correctly rounded on the device as on the host, so one thread per (cluster, dimension) that
/// walks its cluster's rows in order reproduces the host's bits. `order` lists the row indices grouped by cluster (group k =
/// [offsets[k], offsets[k+1]), ascending inside a group: the caller's stable partition of the per-row draws). The loads do
/// not depend on the chain, so 8 of them are in flight per thread.
__global__ __launch_bounds__(128) void random_partition_kernel(const double* __restrict__ xt, size_t ldx, int d,
                                                                const uint32_t* __restrict__ order,
                                                                const uint32_t* __restrict__ offsets, double* __restrict__ means,
                                                                double* __restrict__ sizes)
{
    const int k = blockIdx.x;
    const uint32_t lo = offsets[k], hi = offsets[k + 1];
    const double count0 = sizes[k];
    double count = count0;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {          // (one trip for d <= 128)
        const double* __restrict__ row = xt + (size_t)j * ldx;
        double c = means[(size_t)k * d + j];
        count = count0;
        uint32_t t = lo;
        for (; t + 8 <= hi; t += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = row[order[t + u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                count += 1.0;
                c += (x[u] - c) / count;
            }
        }
        for (; t < hi; ++t) {
            count += 1.0;
            c += (row[order[t]] - c) / count;
        }
        means[(size_t)k * d + j] = c;
    }
    __syncthreads();                      // every thread of the block has read sizes[k]
    if (threadIdx.x == 0) sizes[k] = count0 + (double)(hi - lo);   // (= the count every chain ended with: integers, exact)
}

}  // namespace

void launch_random_partition(const double* xt, size_t ldx, int d, int K, const uint32_t* order, const uint32_t* offsets,
                             double* means, double* sizes, hipStream_t stream)
{
    hipLaunchKernelGGL(random_partition_kernel, dim3(K), dim3(d <= 64 ? 64 : 128), 0, stream, xt, ldx, d, order, offsets, means, sizes);
}

void launch_transpose_to_dim_major(const double* src, int64_t lds, int d, uint64_t n, double* dst, size_t ldd,
                                   uint64_t i0, hipStream_t stream)
{
    if (n == 0) return;
    const unsigned blocks = (unsigned)((n + 63) / 64);
    const int dc = d <= 128 ? d : 128;                          // dimensions per tile: at most 66 KB of LDS
    hipLaunchKernelGGL(transpose_kernel, dim3(blocks, (d + dc - 1) / dc), dim3(256), sizeof(double) * 64 * (dc + 1), stream, src, lds, d,
                       n, dst, ldd, i0, dc);
}

void launch_column_maxabs(const double* xt, size_t ldx, int d, uint64_t n, double* scratch, double* maxabs, hipStream_t stream)
{
    const int parts = 1024;
    hipLaunchKernelGGL(colmax_stage1, dim3(parts, d), dim3(256), 0, stream, xt, ldx, n, scratch);
    hipLaunchKernelGGL(colmax_stage2, dim3(d), dim3(256), 0, stream, scratch, parts, maxabs);
}

void launch_column_sums(const double* xt, size_t ldx, int d, uint64_t n, double* scratch, double* sums, hipStream_t stream)
{
    const int parts = 1024;
    hipLaunchKernelGGL(colsum_stage1, dim3(parts, d), dim3(256), 0, stream, xt, ldx, n, scratch);
    hipLaunchKernelGGL(colsum_stage2, dim3(d), dim3(256), 0, stream, scratch, parts, sums);
}

}  // namespace mlhip
