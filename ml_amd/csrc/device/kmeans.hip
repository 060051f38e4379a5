// K-means assignment + update statistics for gfx950 -- replaces KMeans::assignment_step / assign_label
// (reference ML/KMeans.cpp:153-178), the sums KMeans::update_step needs (:180-192) and the identical
// nearest-centroid loops of the initialisers (ML/Clustering.cpp:44-51, 77-88).
//
// One lane owns one sample (coordinates in VGPRs, X dimension-major in HBM -> coalesced loads); the
// centroid of the current cluster is wave-uniform and arrives through scalar loads. Distances are the
// direct form sum_j (x_j - c_j)^2 accumulated in ascending j as s = fma(t, t, s) -- exactly the IEEE
// operations the host-side assign_label performs (std::fma), so the per-sample distance and the argmin (strict '<', scan from k = 0, first minimum wins, label 0 default)
// are bit-identical to the host point query. The expanded |x|^2 - 2x.c + |c|^2 form is deliberately not
// used: its cancellation can flip labels of nearly equidistant samples.
//
// Update statistics (per-cluster coordinate sums and counts) are accumulated EXACTLY: every coordinate is scaled by a
// per-dimension power of two (chosen at upload from max|x_j| so that |t| < 2^94), cut into three 32-bit limbs and added
// with 64-bit INTEGER atomics (ds_add_u64 on workgroup-private LDS accumulators, or global atomics when K*(3d+1)
// words do not fit). Integer addition is associative, so the sums do not depend on the order in which lanes, waves,
// workgroups or GPUs contribute: the update step is bitwise reproducible run to run (floating-point atomics are not),
// which the reference's own test relies on (same seed => same fit => inertia of 3 initialisations <= inertia of the
// first, Tests/test_KMeans.cpp:75-79). A limb sum cannot overflow: 2^32 samples x 2^32 per limb < 2^64.
#include <cstdlib>

#include "device.hpp"
#include "exact_sum.hpp"

namespace mlhip {
namespace {

// 1024-thread workgroups: the workgroup-private LDS accumulators (K*(3d+1) words, 51 KB at K=256, d=8) would otherwise
// cap the CU at 3 small workgroups = 3 waves per SIMD, too few to cover the dependent FMA chain of a distance.
constexpr int BS = 1024;
// 32 < d <= 64: the coordinates alone take 2d VGPRs, so the workgroup shrinks to 512 threads (256 VGPRs per lane).
constexpr int BS_BIG = 512;

template <int D, bool USE_LDS, int BS>
__global__ __launch_bounds__(BS) void kmeans_assign_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const double* __restrict__ cent, int K,
    const double* __restrict__ scale, uint32_t* __restrict__ labels, const uint32_t* __restrict__ old_labels,
    int have_old, double* __restrict__ min_dist, int accumulate, double* __restrict__ partials, size_t pstride, int copies)
{
    // [copies][K][3d+1] when USE_LDS. With few clusters most lanes of a wave add to the SAME few rows and the LDS atomics serialise
    // (K = 3: ~21 lanes per address); `copies` (a power of two, as many as fit) tables indexed by the low lane bits spread them --
    // integer sums: the copies add up exactly, in any order.
    extern __shared__ u64 acc_lds[];
    __shared__ double red[2 * (BS / 64)];
    const int tid = threadIdx.x;
    const int W = 3 * d + 1;
    double* my_part = partials + (size_t)blockIdx.x * pstride;   // [inertia, changed, K*(3d+1) integer words]
    u64* my_words = reinterpret_cast<u64*>(my_part + 2);
    if (accumulate) {
        if (USE_LDS) {
            for (int e = tid; e < copies * K * W; e += BS) acc_lds[e] = 0;
        } else {
            for (int e = tid; e < K * W; e += BS) my_words[e] = 0;
        }
        __syncthreads();
    }
    double sc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) sc[j] = j < d ? scale[j] : 0.0;

    double inertia = 0.0, changed = 0.0;
    for (uint32_t i = blockIdx.x * (uint32_t)BS + tid; i < n; i += gridDim.x * (uint32_t)BS) {
        double x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = xt[(size_t)j * ldx + i];
        double best = __builtin_inf();
        uint32_t arg = 0;
        // The centroid of cluster k is wave-uniform and arrives through scalar loads; the one of cluster k+1 is
        // requested before the distance to k is computed (register double buffer for D <= 16, where 2*D doubles fit
        // the SGPR file), so the scalar-memory latency is off the critical path.
        if constexpr (D <= 16) {
            double cn[D];
#pragma unroll
            for (int j = 0; j < D; ++j) cn[j] = cent[j];
            for (int k = 0; k < K; ++k) {
                double cc[D];
#pragma unroll
                for (int j = 0; j < D; ++j) cc[j] = cn[j];
                const double* __restrict__ nx = cent + (size_t)(k + 1 < K ? k + 1 : k) * D;
#pragma unroll
                for (int j = 0; j < D; ++j) cn[j] = nx[j];
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const double t = x[j] - cc[j];
                    s = __builtin_fma(t, t, s);
                }
                if (s < best) { best = s; arg = (uint32_t)k; }
            }
        } else {
            for (int k = 0; k < K; ++k) {
                const double* __restrict__ c = cent + (size_t)k * D;
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const double t = x[j] - c[j];
                    s = __builtin_fma(t, t, s);
                }
                if (s < best) { best = s; arg = (uint32_t)k; }
            }
        }
        labels[i] = arg;
        if (min_dist) min_dist[i] = best;
        inertia += best;
        changed += (!have_old || old_labels[i] != arg) ? 1.0 : 0.0;
        if (accumulate) {
            u64* row = (USE_LDS ? acc_lds + (size_t)(tid & (copies - 1)) * K * W : my_words) + (size_t)arg * W;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (j < d) {
                    u64 w0, w1, w2;
                    split_limbs(x[j] * sc[j], w0, w1, w2);
                    atomicAdd(row + 3 * j, w0);
                    atomicAdd(row + 3 * j + 1, w1);
                    atomicAdd(row + 3 * j + 2, w2);
                }
            }
            atomicAdd(row + 3 * d, (u64)1);
        }
    }
    // block sums of inertia / changed (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        inertia += __shfl_down(inertia, off, 64);
        changed += __shfl_down(changed, off, 64);
    }
    constexpr int NWV = BS / 64;
    if ((tid & 63) == 0) {
        red[tid >> 6] = inertia;
        red[NWV + (tid >> 6)] = changed;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < NWV; ++w) { a += red[w]; b += red[NWV + w]; }   // fixed order
        my_part[0] = a;
        my_part[1] = b;
    }
    if (accumulate && USE_LDS) {
        for (int e = tid; e < K * W; e += BS) {
            u64 v = acc_lds[e];
            for (int c = 1; c < copies; ++c) v += acc_lds[(size_t)c * K * W + e];
            my_words[e] = v;
        }
    }
}

/// Update sums as a second sweep, for shapes whose K*(3d+1) accumulator words do not fit LDS next to the assignment
/// kernel's own state. The accumulators are cut into chunks that do fit -- by DIMENSION first: a chunk holds the limbs of
/// DC dimensions for all clusters, so every pass reads its DC rows of X densely (all lanes active, coalesced; X is read
/// exactly once over all passes) plus the labels; only when even one dimension of all K clusters is too much (K > ~3000)
/// are the clusters chunked as well (KC < K: lanes outside the chunk idle). Same exact limb sums as the fused path
/// (ds_add_u64), flushed per chunk into the workgroup's partial block -- instead of global-memory atomics, which serialise
/// on popular clusters (N=12.5M, d=16, K=256: 25 ms; this sweep: 0.7 ms on top of the 2.2 ms assignment).
constexpr int BSU = 1024;
__global__ __launch_bounds__(BSU) void kmeans_update_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n, int d, const uint32_t* __restrict__ labels,
    const double* __restrict__ scale, int K, int KC, int DC, double* __restrict__ partials, size_t pstride)
{
    extern __shared__ u64 acc_lds[];      // [KC][3*DC + 1]
    const int tid = threadIdx.x;
    const int W = 3 * d + 1;              // words per cluster in the partial block
    const int WC = 3 * DC + 1;            // words per cluster in LDS (last one: the count, used by the first dim chunk)
    u64* my_words = reinterpret_cast<u64*>(partials + (size_t)blockIdx.x * pstride + 2);
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = min(KC, K - k0);
        for (int j0 = 0; j0 < d; j0 += DC) {
            const int dc = min(DC, d - j0);
            for (int e = tid; e < kc * WC; e += BSU) acc_lds[e] = 0;
            __syncthreads();
            for (uint32_t i = blockIdx.x * (uint32_t)BSU + tid; i < n; i += gridDim.x * (uint32_t)BSU) {
                const uint32_t rel = labels[i] - (uint32_t)k0;
                if (rel < (uint32_t)kc) {
                    u64* row = acc_lds + (size_t)rel * WC;
                    for (int j = 0; j < dc; ++j) {
                        u64 w0, w1, w2;
                        split_limbs(xt[(size_t)(j0 + j) * ldx + i] * scale[j0 + j], w0, w1, w2);
                        atomicAdd(row + 3 * j, w0);
                        atomicAdd(row + 3 * j + 1, w1);
                        atomicAdd(row + 3 * j + 2, w2);
                    }
                    if (j0 == 0) atomicAdd(row + 3 * DC, (u64)1);
                }
            }
            __syncthreads();
            for (int e = tid; e < kc * 3 * dc; e += BSU) {
                const int k = e / (3 * dc), w = e - k * 3 * dc;
                my_words[(size_t)(k0 + k) * W + 3 * j0 + w] = acc_lds[(size_t)k * WC + w];
            }
            if (j0 == 0)
                for (int k = tid; k < kc; k += BSU) my_words[(size_t)(k0 + k) * W + 3 * d] = acc_lds[(size_t)k * WC + 3 * DC];
            __syncthreads();
        }
    }
}

/// One output element of [inertia, n_changed, counts(K), sums(K*d)] by one wave, the lanes striding over the per-workgroup partial blocks
/// four at a time (round 5: the loads of a trip in flight together -- 11.8 -> ... us at K = 256, d = 8 over 512 blocks). inertia / changed:
/// lane-strided partial sums combined by a shuffle tree (a fixed order for a given number of partials); counts and coordinate sums:
/// exact integer sums of the limb words (order-free), converted to double once.
__device__ __forceinline__ void kmeans_reduce_element(const double* __restrict__ partials, int n_blocks, size_t pstride, int K, int d,
                                                      const double* __restrict__ scale, double* out, int e, int total, int lane)
{
    auto wave_sum = [](u64 v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
    };
    const int W = 3 * d + 1;
    if (e < 2) {
        double v = 0.0;
        for (int b = lane; b < n_blocks; b += 64) v += partials[(size_t)b * pstride + e];   // (the order of kmeans_reduce_kernel: same bits)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) out[e] = v;
    } else if (e < total) {
        const int k = (e - 2) / (d + 1), j = (e - 2) - k * (d + 1);
        const u64* words = reinterpret_cast<const u64*>(partials + 2) + (size_t)k * W;
        if (j == d) {
            u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            int b = lane;
            for (; b + 192 < n_blocks; b += 256) {
                c0 += words[(size_t)b * pstride + 3 * d];
                c1 += words[(size_t)(b + 64) * pstride + 3 * d];
                c2 += words[(size_t)(b + 128) * pstride + 3 * d];
                c3 += words[(size_t)(b + 192) * pstride + 3 * d];
            }
            for (; b < n_blocks; b += 64) c0 += words[(size_t)b * pstride + 3 * d];
            const u64 c = wave_sum((c0 + c1) + (c2 + c3));                                    // (integer sums: order-free)
            if (lane == 0) out[2 + k] = (double)c;
        } else {
            u64 w0 = 0, w1 = 0, w2 = 0;
            int b = lane;
            for (; b + 192 < n_blocks; b += 256) {
                const u64* p0 = words + (size_t)b * pstride + 3 * j;
                const u64* p1 = p0 + 64 * pstride;
                const u64* p2 = p1 + 64 * pstride;
                const u64* p3 = p2 + 64 * pstride;
                const u64 a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
                const u64 c0 = p2[0], c1 = p2[1], c2 = p2[2], d0 = p3[0], d1 = p3[1], d2 = p3[2];
                w0 += (a0 + b0) + (c0 + d0);
                w1 += (a1 + b1) + (c1 + d1);
                w2 += (a2 + b2) + (c2 + d2);
            }
            for (; b < n_blocks; b += 64) {
                const u64* p = words + (size_t)b * pstride + 3 * j;
                w0 += p[0];
                w1 += p[1];
                w2 += p[2];
            }
            w0 = wave_sum(w0); w1 = wave_sum(w1); w2 = wave_sum(w2);
            if (lane == 0) {                                                                  // (the conversion of kmeans_reduce_kernel)
                w1 += w0 >> 32;  w0 &= 0xffffffffull;
                const long long top = (long long)w2 + (long long)(w1 >> 32);
                w1 &= 0xffffffffull;
                const double v = __builtin_fma((double)top, 0x1p64, __builtin_fma((double)w1, 0x1p32, (double)w0));
                out[2 + K + (size_t)k * d + j] = v / scale[j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void kmeans_reduce_kernel(const double* __restrict__ partials, int n_blocks, size_t pstride,
                                                             int K, int d, int accumulate, const double* __restrict__ scale,
                                                             double* __restrict__ out)
{
    const int total = 2 + (accumulate ? K * (d + 1) : 0);
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);        // wave-uniform
    kmeans_reduce_element(partials, n_blocks, pstride, K, d, scale, out, e, total, threadIdx.x & 63);
}

template <int D>
void launch_t(const KmeansArgs& a, int grid, int use_lds, size_t pstride, hipStream_t stream)
{
    constexpr int BSZ = D <= kRegDim ? BS : BS_BIG;
    const size_t table = sizeof(u64) * (size_t)a.K * (3 * a.d + 1);
    int copies = 1;                                              // as many accumulator tables as fit 64 KB, at most 8
    while (use_lds && copies < 8 && 2 * copies * table <= 64 * 1024) copies *= 2;
    const size_t smem = use_lds ? copies * table : 0;
    if (use_lds)
        hipLaunchKernelGGL((kmeans_assign_kernel<D, true, BSZ>), dim3(grid), dim3(BSZ), smem, stream, a.xt, a.ldx, a.n, a.d,
                           a.centroids, a.K, a.scale, a.labels, a.old_labels, a.have_old, a.min_dist, a.accumulate, a.partials, pstride, copies);
    else
        hipLaunchKernelGGL((kmeans_assign_kernel<D, false, BSZ>), dim3(grid), dim3(BSZ), smem, stream, a.xt, a.ldx, a.n, a.d,
                           a.centroids, a.K, a.scale, a.labels, a.old_labels, a.have_old, a.min_dist, a.accumulate, a.partials, pstride, 1);
}

inline int kmeans_grid(int num_cus) { return num_cus * 2; }   // two 1024-thread workgroups (32 waves) per CU

}  // namespace

size_t kmeans_scratch_doubles(int d, int K, int num_cus)
{
    return (size_t)kmeans_grid(num_cus) * (2 + (size_t)K * (3 * d + 1));
}

/// Separate update sweep (see kmeans_update_kernel) over the same `grid` partial blocks the assignment kernel used.
void launch_kmeans_update(const KmeansArgs& a, int grid, size_t pstride, hipStream_t stream)
{
    const size_t budget = 72 * 1024 / sizeof(u64);           // words; two 1024-thread workgroups per CU
    // as many whole dimensions of all K clusters as fit; if not even one does, chunk the clusters too
    const size_t per_cluster = budget / (size_t)a.K;          // words available per cluster when all K are resident
    int KC = a.K, DC = per_cluster >= 4 ? (int)((per_cluster - 1) / 3) : 0;
    if (DC > a.d) DC = a.d;
    if (DC < 1) {
        DC = 1;
        KC = (int)(budget / 4);
        if (KC < 1) KC = 1;
    }
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(grid), dim3(BSU), sizeof(u64) * (size_t)KC * (3 * DC + 1), stream, a.xt, a.ldx,
                       a.n, a.d, a.labels, a.scale, a.K, KC, DC, a.partials, pstride);
}

bool kmeans_mfma_supported(int D, int K);
int launch_kmeans_mfma(const KmeansArgs& a, int num_cus, hipStream_t stream);

int launch_kmeans_assign(const KmeansArgs& a_in, int num_cus, hipStream_t stream)
{
    // Matrix-core search with exact recheck (kmeans_mfma.hip) where it applies; MLHIP_KMEANS=valu forces this file's kernel,
    // MLHIP_KMEANS=mfma the matrix-core one wherever it exists.
    const char* e = std::getenv("MLHIP_KMEANS");
    const bool force_valu = e && e[0] == 'v', force_mfma = e && e[0] == 'm';
    // Few clusters (K <= 16, d <= 32): the direct-form kernel of this file -- the matrix-core search pads K to 16 rows and pays its
    // exact recheck per sample whatever K is (round 4, N = 10M: d = 4, K = 8 0.230 -> 0.155 ms, K = 16 0.212 -> 0.164; d = 8, K = 16
    // 0.318 -> 0.276; d = 16, K = 8 0.595 -> 0.515; d = 32, K = 16 0.486 -> 0.474; K = 32: the matrix cores win again). Labels and
    // distances are the same bits either way.
    // From 2^21 samples on: below, this kernel's 1024-thread workgroups (zeroing and flushing their LDS accumulators) are the
    // slower ones (N = 20k, d = 32, K = 4: 69 against 52 us per step; N = 1M, d = 16, K = 16: 87 against 83).
    const bool few = a_in.K <= (a_in.D <= 8 ? 24 : 16) && a_in.D <= kRegDim && a_in.n >= (1u << 21) && !force_mfma;   // (K = 24: d = 4 -14 %, d = 8 -7 %)
    // (above d = 64 only the matrix-core kernel exists)
    if ((!(force_valu || few) || a_in.D > kMidDim) && kmeans_mfma_supported(a_in.D, a_in.K)) return launch_kmeans_mfma(a_in, num_cus, stream);
    const size_t pstride = 2 + (size_t)a_in.K * (3 * a_in.d + 1);
    if (a_in.D > kMaxDim) {
        // big_dim.hip (register-blocked) or generic_dim.hip (plain): exact assignment, update sums by the separate sweep
        if (big_dim_kmeans_applies(a_in.D) && (size_t)kmeans_grid(num_cus) * pstride <= a_in.partials_capacity) {
            const int used = launch_kmeans_assign_big(a_in, kmeans_grid(num_cus), pstride, stream);
            if (used > 0) {
                if (a_in.accumulate) launch_kmeans_update(a_in, used, pstride, stream);
                return used;
            }
        }
        int grid = kmeans_grid(num_cus);
        const uint32_t need = (a_in.n + 255) / 256;
        if ((uint32_t)grid > need) grid = (int)(need ? need : 1);
        if ((size_t)grid * pstride > a_in.partials_capacity) return -2;
        launch_kmeans_assign_generic(a_in, grid, pstride, stream);
        if (a_in.accumulate) launch_kmeans_update(a_in, grid, pstride, stream);
        return grid;
    }
    int grid = kmeans_grid(num_cus);
    const uint32_t bs = a_in.D <= kRegDim ? BS : BS_BIG;
    const uint32_t blocks_needed = (a_in.n + bs - 1) / bs;
    if ((uint32_t)grid > blocks_needed) grid = (int)(blocks_needed ? blocks_needed : 1);
    if ((size_t)grid * pstride > a_in.partials_capacity) return -2;
    // Accumulators that fit LDS are updated by the assignment kernel itself; otherwise by a separate sweep afterwards.
    const bool fits = (size_t)a_in.K * (3 * a_in.d + 1) * sizeof(u64) <= 64 * 1024;
    const int use_lds = fits ? 1 : 0;
    KmeansArgs a = a_in;
    if (!fits) a.accumulate = 0;
    switch (a.D) {
    case 1: launch_t<1>(a, grid, use_lds, pstride, stream); break;
    case 2: launch_t<2>(a, grid, use_lds, pstride, stream); break;
    case 3: launch_t<3>(a, grid, use_lds, pstride, stream); break;
    case 4: launch_t<4>(a, grid, use_lds, pstride, stream); break;
    case 6: launch_t<6>(a, grid, use_lds, pstride, stream); break;
    case 8: launch_t<8>(a, grid, use_lds, pstride, stream); break;
    case 12: launch_t<12>(a, grid, use_lds, pstride, stream); break;
    case 16: launch_t<16>(a, grid, use_lds, pstride, stream); break;
    case 20: launch_t<20>(a, grid, use_lds, pstride, stream); break;
    case 24: launch_t<24>(a, grid, use_lds, pstride, stream); break;
    case 28: launch_t<28>(a, grid, use_lds, pstride, stream); break;
    case 32: launch_t<32>(a, grid, use_lds, pstride, stream); break;
    case 40: launch_t<40>(a, grid, use_lds, pstride, stream); break;
    case 48: launch_t<48>(a, grid, use_lds, pstride, stream); break;
    case 56: launch_t<56>(a, grid, use_lds, pstride, stream); break;
    case 64: launch_t<64>(a, grid, use_lds, pstride, stream); break;
    default: return -1;
    }
    if (a_in.accumulate && !fits) launch_kmeans_update(a_in, grid, pstride, stream);
    return grid;
}

namespace {
/// One thread per (cluster, padded coordinate): the same IEEE division as the host's closing loop (mlhip_kmeans_step).
__global__ __launch_bounds__(256) void kmeans_close_kernel(double* __restrict__ out, int K, int d, int D, double* __restrict__ next,
                                                            double* __restrict__ mirror)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (mirror && e < 2 + K) mirror[e] = out[e];          // inertia, changed, counts: as they are
    if (e >= K * D) return;
    const int k = e / D, j = e - k * D;
    double v = 0.0;
    if (j < d) {
        const double c = out[2 + k];
        double* sum = out + 2 + K + (size_t)k * d + j;
        v = c > 0 ? *sum / c : 0.0;
        *sum = v;
        if (mirror) mirror[2 + K + (size_t)k * d + j] = v;
    }
    next[e] = v;
}
}  // namespace

#ifdef MLHIP_EXPERIMENTS
namespace {
/// kmeans_reduce_kernel and kmeans_close_kernel in ONE launch (single rank: no all-reduce between the two; round 5): every workgroup
/// reduces its four output elements as before, takes a ticket, and the workgroup that draws the last one -- every element of `out` is
/// then written and, behind the fences, visible -- forms the means, the next centroid table and the pinned mirror. One dependent
/// dispatch less per step -- and measured SLOWER: 32.2 us for the one launch against 11.8 + 4.2 us for the two (K = 256, d = 8, 512
/// partial blocks: the last workgroup closes 2 048 entries and writes the 18 KB mirror alone, behind the slowest of 577 reductions),
/// 1.094 against 1.074 ms per step at N = 12.5M. `make EXPERIMENTS=1` + MLHIP_KMEANS_FUSED=1 only (profiles/r05_kmeans_step.txt).
__global__ __launch_bounds__(256) void kmeans_reduce_close_kernel(const double* __restrict__ partials, int n_blocks, size_t pstride, int K, int d,
                                                                   int D, const double* __restrict__ scale, double* out, double* __restrict__ next,
                                                                   double* __restrict__ mirror, unsigned* __restrict__ ticket, unsigned ticket_base)
{
    __shared__ int is_last;
    const int total = 2 + K * (d + 1);
    kmeans_reduce_element(partials, n_blocks, pstride, K, d, scale, out, blockIdx.x * 4 + (threadIdx.x >> 6), total, threadIdx.x & 63);
    // The in-launch combine of the gfx950 guide (counter form of its hand-off): every wave's stores drained, ONE agent-scope release
    // and ONE relaxed ticket per workgroup; the workgroup that draws the launch's last ticket acquires once and reads with plain loads.
    // The counter is never reset: launch number m of a handle owns the tickets ticket_base .. ticket_base + gridDim.x - 1.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = t - ticket_base == gridDim.x - 1 ? 1 : 0;
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!is_last) return;
    const int threads = K * D > K + 2 ? K * D : K + 2;
    for (int t = threadIdx.x; t < threads; t += 256) {        // kmeans_close_kernel's body
        if (mirror && t < 2 + K) mirror[t] = out[t];
        if (t >= K * D) continue;
        const int k = t / D, j = t - k * D;
        double v = 0.0;
        if (j < d) {
            const double c = out[2 + k];
            double* sum = out + 2 + K + (size_t)k * d + j;
            v = c > 0 ? *sum / c : 0.0;
            *sum = v;
            if (mirror) mirror[2 + K + (size_t)k * d + j] = v;
        }
        next[t] = v;
    }
}
}  // namespace

unsigned launch_kmeans_reduce_close(const KmeansArgs& a, int n_partials, int D, double* next, double* mirror, unsigned* ticket,
                                    unsigned ticket_base, hipStream_t stream)
{
    const size_t pstride = 2 + (size_t)a.K * (3 * a.d + 1);
    const int total = 2 + a.K * (a.d + 1);
    const unsigned grid = (unsigned)((total + 3) / 4);
    hipLaunchKernelGGL(kmeans_reduce_close_kernel, dim3(grid), dim3(256), 0, stream, a.partials, n_partials, pstride, a.K, a.d, D, a.scale, a.out,
                       next, mirror, ticket, ticket_base);
    return grid;
}

#endif

void launch_kmeans_close(double* out, int K, int d, int D, double* next, double* mirror, hipStream_t stream)
{
    const int threads = K * D > K + 2 ? K * D : K + 2;
    hipLaunchKernelGGL(kmeans_close_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, out, K, d, D, next, mirror);
}

void launch_kmeans_reduce(const KmeansArgs& a, int n_partials, hipStream_t stream)
{
    const size_t pstride = 2 + (size_t)a.K * (3 * a.d + 1);
    const int total = 2 + (a.accumulate ? a.K * (a.d + 1) : 0);
    hipLaunchKernelGGL(kmeans_reduce_kernel, dim3((total + 3) / 4), dim3(256), 0, stream, a.partials, n_partials, pstride,
                       a.K, a.d, a.accumulate, a.scale, a.out);
}

}  // namespace mlhip
