// Data-movement kernels: sample-major (the caller's d x N column-major block, reference
// ML/Clustering.hpp:28-33) -> dimension-major HBM layout, and deterministic column sums.
#include "device.hpp"

#include <algorithm>

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

// ---- K-means++ draw on the device (KPP::init, reference ML/Clustering.cpp:39-59) ----------------------------------------------
// std::discrete_distribution turns the N weights into p_i = w_i / sum (sum: a sequential floating-point sum), their sequential
// cumulative sums cp_i, and returns the first i with cp_i >= u (the last one forced to 1). The sequential sums cannot be
// reproduced in parallel bit for bit -- but the INDEX can be certified: every quantity involved is a sum of non-negative terms
// bounded by 1, so |cp_i - c~_i| <= delta = (4 N + 16384) 2^-53 for the tree-summed c~_i formed here, whatever the order. With
// lo = first i with c~_i >= u - delta and hi = first i with c~_i >= u + delta the reference's index lies in [lo, hi]; lo == hi
// (all but ~2 delta N of the draws) settles it, otherwise the caller evaluates the reference's sequential sums on the host.
constexpr int kKppChunk = 4096;   // weights per workgroup

/// weights = first ? dist : min(weights, dist); block sums of the new weights (fixed-order tree).
__global__ __launch_bounds__(256) void kpp_update_kernel(double* __restrict__ weights, const double* __restrict__ dist, uint32_t n,
                                                          int first, double* __restrict__ bsum)
{
    __shared__ double red[256];
    const uint32_t base = blockIdx.x * (uint32_t)kKppChunk;
    double acc = 0.0;
#pragma unroll 4
    for (int t = 0; t < kKppChunk / 256; ++t) {
        const uint32_t i = base + t * 256u + threadIdx.x;
        if (i < n) {
            const double dv = dist[i];
            const double w = first ? dv : fmin(weights[i], dv);
            weights[i] = w;
            acc += w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}

/// Exclusive prefix of the block sums (one workgroup, 1024 sums per trip) and the total: out[0] = sum, out[1] = lo, out[2] = hi
/// (the latter two initialised to the last row of the WHOLE sample, as doubles: exact below 2^53).
__global__ __launch_bounds__(1024) void kpp_scan_kernel(const double* __restrict__ bsum, int nb, double* __restrict__ boff,
                                                         double default_index, double* __restrict__ out)
{
    __shared__ double buf[1024];
    double carry = 0.0;
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const double v = b < nb ? bsum[b] : 0.0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                     // inclusive Hillis-Steele scan
            const double add = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0.0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (b < nb) boff[b] = carry + (buf[threadIdx.x] - v);
        const double total = buf[1023];
        __syncthreads();
        carry += total;
    }
    if (threadIdx.x == 0) {
        out[0] = carry;
        out[1] = out[2] = default_index;
    }
}

__device__ __forceinline__ void atomic_min_index(double* slot, uint64_t i)
{
    // indices are stored as doubles (exact); their bit patterns order like the values for non-negative doubles
    atomicMin(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong((double)i));
}

/// lo / hi over the rows of this block (global indices row0 + i; the last row of the whole sample is the default and never a
/// candidate). `offset`: the weight sum of the ranks before this one, `total`: that of all ranks (single rank: 0 and out[0]);
/// u and delta are scaled by the total, so there is no division per row.
__global__ __launch_bounds__(256) void kpp_find_kernel(const double* __restrict__ weights, uint32_t n, const double* __restrict__ bsum,
                                                        const double* __restrict__ boff, double offset, double total, double u, double delta,
                                                        uint64_t row0, uint64_t n_global, double* __restrict__ out)
{
    __shared__ double pre[256];
    const double t_lo = (u - delta) * total, t_hi = (u + delta) * total;       // thresholds on the unnormalised prefix sums
    const double slack = 4.0 * kKppChunk * 0x1p-53 * total;
    const uint32_t base = blockIdx.x * (uint32_t)kKppChunk;
    const double b_lo = offset + boff[blockIdx.x], b_hi = b_lo + bsum[blockIdx.x];
    if (base >= n || row0 + base + 1 >= n_global) return;                      // only the forced last row (or nothing) here
    if (b_hi + slack < t_lo) return;                                           // every prefix of this chunk is below both thresholds
    if (b_lo - slack > t_hi) {                                                 // every prefix is above both: the chunk's first row
        if (threadIdx.x == 0) { atomic_min_index(out + 1, row0 + base); atomic_min_index(out + 2, row0 + base); }
        return;
    }
    // 16 consecutive rows per thread; exclusive prefix over the 256 thread sums, then every thread walks its rows
    constexpr int PER = kKppChunk / 256;
    const uint32_t first = base + threadIdx.x * PER;
    double w[PER], acc = 0.0;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        w[t] = first + t < n ? weights[first + t] : 0.0;
        acc += w[t];
    }
    pre[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const double add = (int)threadIdx.x >= off ? pre[threadIdx.x - off] : 0.0;
        __syncthreads();
        pre[threadIdx.x] += add;
        __syncthreads();
    }
    double c = b_lo + (pre[threadIdx.x] - acc);
    bool seen_lo = false;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const uint32_t i = first + t;
        c += w[t];
        if (i < n && row0 + i + 1 < n_global) {                                // (the last row of the sample is the default)
            if (!seen_lo && c >= t_lo) { atomic_min_index(out + 1, row0 + i); seen_lo = true; }
            if (c >= t_hi) { atomic_min_index(out + 2, row0 + i); break; }
        }
    }
}


// The device group's in-process all-reduce (runtime/group.cpp): out[i] = ((slot_0[i] + slot_1[i]) + slot_2[i]) + ... in shard order --
// every shard runs this very sum on the same inputs, so all hold bit-identical results (slots of other devices: peer access).
__global__ __launch_bounds__(256) void group_sum_kernel(GroupSumSlots slots, int n, double* __restrict__ out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        double s = slots.p[0][i];
        for (int r = 1; r < n; ++r) s += slots.p[r][i];
        out[i] = s;
    }
}

}  // namespace

int kpp_blocks(uint32_t n) { return (int)((n + kKppChunk - 1) / kKppChunk); }

void launch_kpp_update(double* weights, const double* dist, uint32_t n, int first, double default_index, double* bsum, double* boff,
                       double* out, hipStream_t stream)
{
    const int nb = kpp_blocks(n);
    if (nb > 0) hipLaunchKernelGGL(kpp_update_kernel, dim3(nb), dim3(256), 0, stream, weights, dist, n, first, bsum);   // (an empty shard: sum 0)
    hipLaunchKernelGGL(kpp_scan_kernel, dim3(1), dim3(1024), 0, stream, bsum, nb, boff, default_index, out);
}

void launch_kpp_find(const double* weights, uint32_t n, const double* bsum, const double* boff, double offset, double total, double u,
                     double delta, uint64_t row0, uint64_t n_global, double* out, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(kpp_find_kernel, dim3(kpp_blocks(n)), dim3(256), 0, stream, weights, n, bsum, boff, offset, total, u, delta, row0,
                       n_global, out);
}

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

void launch_group_sum(const GroupSumSlots& slots, int n, double* out, size_t count, hipStream_t stream)
{
    if (!count) return;
    const unsigned grid = (unsigned)std::min<size_t>((count + 255) / 256, 256);
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid), dim3(256), 0, stream, slots, n, out, count);
}

}  // namespace mlhip
