// The WHOLE step loop of KMeans::fit_once (reference ML/KMeans.cpp:80-110: assignment_step, update_step, the "same labels twice" and
// "centroids moved less than the tolerance" tests) in ONE launch of ONE workgroup, for the reference's own benchmark regime
// (Benchmarks/bm_KMeans.cpp: d = 2, K = 3, N = 100 ... 100 000, three initialisations per fit). As three dependent launches and a
// read-back per step (assignment kernel, reduction, closing: runtime/kmeans.cpp km_iterate) such a step costs 21 us whatever N is; here a
// step of a block that one workgroup holds costs 7 - 10 us (tools/kmeans_resident_time.py).
//
// With at most 4 096 samples (2-d; 1 024 in 6 dimensions) one workgroup of 1024 threads holds the whole problem IN REGISTERS: the centroid table, the exact accumulator words and
// the stopping tests live in LDS, the steps are separated by workgroup barriers -- no exchange between workgroups, nothing to wait for
// but this workgroup's own waves.
//
// BIT-identical to the three-launch loop (tests/test_gpu_kmeans_resident.py), inertia included: the direct-form distances with the
// same fma chain and strict '<' (kmeans.hip kmeans_assign_kernel -- the kernel the launches use for D = 1, 2, 3, 6), the same exact
// limb sums (integer: order-free) and their conversion (kmeans_reduce_element), the same division (kmeans_close_kernel), the host's
// shift test (km_iterate), and the inertia summed in the ORDER of the launches: thread t of the launches' workgroup b holds sample
// 1024 b + t -- here thread t walks b = 0, 1, ... itself, every "workgroup" sum is the same shuffle tree per wave and the same
// wave-by-wave addition, and the sums of the "workgroups" are added as kmeans_reduce_element adds the partial blocks.
#include "device.hpp"
#include "exact_sum.hpp"

#pragma clang fp contract(off)     // (the distances use explicit fma; everything else is the launches' / the host's statement-by-statement arithmetic)

namespace mlhip {
namespace {

constexpr int RBS = 1024;                  // threads (the launches' workgroup size for D <= 32)
constexpr int RMAXB = kKmResidentMaxN / RBS;   // "workgroups" of the launches this one stands for

/// NB: samples per thread (sample 1024 b + t in slot b). They are loaded ONCE and stay in registers for the whole loop, as do the labels
/// of the assignment before -- a step touches memory only to store labels and distances. (Re-reading the samples every step, a slot at
/// a time, costs a memory round trip per slot and step.)
template <int D, int NB>
__global__ __launch_bounds__(RBS) void kmeans_resident_kernel(KmResidentArgs a)
{
    extern __shared__ u64 acc_lds[];                       // [copies][K][3d+1]
    __shared__ double cur[kKmResidentMaxK * D], old[kKmResidentMaxK * D], upd[kKmResidentMaxK * D];
    __shared__ double cnt[kKmResidentMaxK], red_w[RMAXB * (RBS / 64)], part[RMAXB];
    __shared__ double s_inertia;
    __shared__ unsigned s_changed;
    __shared__ int s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, d = a.d, W = 3 * d + 1, copies = a.copies;
    const uint32_t n = a.n;
    const int nblk = (int)((n + RBS - 1) / RBS);
    double sc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) sc[j] = j < d ? a.scale[j] : 0.0;
    for (int e = tid; e < K * D; e += RBS) {
        cur[e] = a.cent[e];
        old[e] = 0.0;
    }
    int lb = a.label_buf;                                   // buffer holding the labels of the assignment before
    bool have_old = a.have_old != 0;
    uint32_t steps = 0;
    int converged = 0;
    double x[NB][D];
    uint32_t lab_prev[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t i = (uint32_t)b * RBS + (uint32_t)tid;
        const uint32_t ic = i < n ? i : n - 1;
#pragma unroll
        for (int j = 0; j < D; ++j) x[b][j] = a.xt[(size_t)j * a.ldx + ic];
        lab_prev[b] = have_old ? a.labels[lb][ic] : 0u;
    }
    __syncthreads();

    // One assignment_step over all samples (+ the update sums): labels, distances, inertia in the launches' order, changed labels.
    auto assignment = [&](bool accumulate) {
        if (accumulate)
            for (int e = tid; e < copies * K * W; e += RBS) acc_lds[e] = 0;
        if (tid == 0) s_changed = 0;
        __syncthreads();
        uint32_t* __restrict__ lab_new = a.labels[lb ^ 1];
        unsigned changed = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nblk) {                                 // (uniform)
            const uint32_t i = (uint32_t)b * RBS + (uint32_t)tid;
            double inertia = 0.0;
            if (i < n) {
                double best = __builtin_inf();
                uint32_t arg = 0;
                for (int k = 0; k < K; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const double t = x[b][j] - cur[k * D + j];
                        s = __builtin_fma(t, t, s);
                    }
                    if (s < best) { best = s; arg = (uint32_t)k; }
                }
                changed += (!have_old || lab_prev[b] != arg) ? 1u : 0u;
                lab_prev[b] = arg;
                lab_new[i] = arg;
                a.min_dist[i] = best;
                inertia += best;
                if (accumulate) {
                    u64* row = acc_lds + (size_t)(tid & (copies - 1)) * K * W + (size_t)arg * W;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        if (j < d) {
                            u64 w0, w1, w2;
                            split_limbs(x[b][j] * sc[j], w0, w1, w2);
                            atomicAdd(row + 3 * j, w0);
                            atomicAdd(row + 3 * j + 1, w1);
                            atomicAdd(row + 3 * j + 2, w2);
                        }
                    }
                    atomicAdd(row + 3 * d, (u64)1);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) inertia += __shfl_down(inertia, off, 64);
            if (lane == 0) red_w[b * (RBS / 64) + wave] = inertia;
            }
        }
        if (changed) atomicAdd(&s_changed, changed);
        __syncthreads();
        if (tid < nblk) {                                   // the launches' workgroup tid: its waves in order
            double v = 0.0;
            for (int w = 0; w < RBS / 64; ++w) v += red_w[tid * (RBS / 64) + w];
            part[tid] = v;
        }
        __syncthreads();
        if (wave == 0) {                                    // kmeans_reduce_element over the partial blocks
            double v = 0.0;
            for (int b = lane; b < nblk; b += 64) v += part[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) s_inertia = v;
        }
        lb ^= 1;
        have_old = true;
        __syncthreads();
    };

    for (uint32_t step = 0; step < a.max_steps; ++step) {
        assignment(true);
        // counts and coordinate sums of the step (kmeans_reduce_element's conversion), the means (kmeans_close_kernel)
        for (int e = tid; e < K * (d + 1); e += RBS) {
            const int k = e / (d + 1), j = e - k * (d + 1);
            if (j == d) {
                u64 c = 0;
                for (int cp = 0; cp < copies; ++cp) c += acc_lds[(size_t)cp * K * W + (size_t)k * W + 3 * d];
                cnt[k] = (double)c;
            }
        }
        __syncthreads();
        ++steps;
        if (step > 0 && s_changed == 0) {                  // same labels twice (ML/KMeans.cpp:84-89): the centroids stay as they are
            converged = 1;
            break;
        }
        for (int e = tid; e < K * D; e += RBS) {
            const int k = e / D, j = e - k * D;
            double v = 0.0;
            if (j < d) {
                u64 w0 = 0, w1 = 0, w2 = 0;
                for (int cp = 0; cp < copies; ++cp) {
                    const u64* p = acc_lds + (size_t)cp * K * W + (size_t)k * W + 3 * j;
                    w0 += p[0];
                    w1 += p[1];
                    w2 += p[2];
                }
                w1 += w0 >> 32;  w0 &= 0xffffffffull;
                const long long top = (long long)w2 + (long long)(w1 >> 32);
                w1 &= 0xffffffffull;
                const double sum = __builtin_fma((double)top, 0x1p64, __builtin_fma((double)w1, 0x1p32, (double)w0)) / a.scale[j];
                const double c = cnt[k];
                v = c > 0 ? sum / c : 0.0;                  // (empty cluster -> origin, ML/KMeans.cpp:184)
            }
            upd[e] = v;
        }
        __syncthreads();
        for (int e = tid; e < K * D; e += RBS) {            // update_step (:180-192)
            old[e] = cur[e];
            cur[e] = upd[e];
        }
        __syncthreads();
        if (step > 0) {
            if (tid == 0) {                                 // the host's loop over the K d coordinates, in its order
                double shift = 0;
                for (int k = 0; k < K; ++k)
                    for (int j = 0; j < d; ++j) {
                        const double delta = cur[k * D + j] - old[k * D + j];
                        shift += delta * delta;
                    }
                s_stop = shift < a.atol ? 1 : 0;
            }
            __syncthreads();
            if (s_stop) {                                   // (:103-108) one more assignment under the final centroids
                assignment(false);
                converged = 1;
                break;
            }
        }
    }
    // results: [steps, converged, inertia, label buffer, counts(K), centroids(K d), old centroids(K d)] -> pinned host memory; the final table
    __syncthreads();
    double* out = a.out;
    if (tid == 0) {
        out[0] = (double)steps;
        out[1] = (double)converged;
        out[2] = s_inertia;
        out[3] = (double)lb;
    }
    for (int k = tid; k < K; k += RBS) out[4 + k] = cnt[k];
    for (int e = tid; e < K * d; e += RBS) {
        const int k = e / d, j = e - k * d;
        out[4 + K + e] = cur[k * D + j];
        out[4 + K + (size_t)K * d + e] = old[k * D + j];
    }
    for (int e = tid; e < K * D; e += RBS) a.cent[e] = cur[e];
}

template <int D, int NB>
bool launch_t(const KmResidentArgs& a_in, hipStream_t stream)
{
    KmResidentArgs a = a_in;
    const size_t table = sizeof(u64) * (size_t)a.K * (3 * a.d + 1);
    int copies = 1;                                         // as the launches: as many accumulator tables as fit, at most 8
    while (copies < 8 && 2 * copies * table <= 32 * 1024) copies *= 2;
    a.copies = copies;
    hipLaunchKernelGGL((kmeans_resident_kernel<D, NB>), dim3(1), dim3(RBS), copies * table, stream, a);
    return true;
}

/// Samples per thread: what the 128 registers of a 1024-thread workgroup hold without scratch -- and about where ONE compute unit's LDS
/// atomics stop paying (7 per sample at d = 2, all in one LDS: 1.7 us per slot and step; the launches spread them over the chip at
/// 21 us per step whatever N): 4 096 samples in one or two dimensions, 2 048 in three, 1 024 in five or six.
constexpr int max_slots(int D) { return D <= 2 ? 4 : (D == 3 ? 2 : 1); }

template <int D>
bool launch_d(const KmResidentArgs& a, hipStream_t stream)
{
    const int nblk = (int)((a.n + RBS - 1) / RBS);
    if (nblk <= 1) return launch_t<D, 1>(a, stream);
    if constexpr (max_slots(D) >= 2) { if (nblk <= 2) return launch_t<D, 2>(a, stream); }
    if constexpr (max_slots(D) >= 4) { if (nblk <= 4) return launch_t<D, 4>(a, stream); }
    return false;
}

}  // namespace

bool kmeans_resident_supported(int D, int d, int K, uint64_t n)
{
    // D = 1, 2, 3, 6: the dimensions whose step the launches run on the direct-form kernel (kmeans.hip) -- 4 and 8 go to the matrix-core
    // search there, whose inertia is summed in another order
    if (!(D == 1 || D == 2 || D == 3 || D == 6) || d < 1 || d > D) return false;
    if (K < 1 || K > kKmResidentMaxK || n < 1 || n > (uint64_t)RBS * (uint64_t)max_slots(D)) return false;
    return sizeof(u64) * (size_t)K * (3 * d + 1) <= 32 * 1024;
}

bool launch_kmeans_resident(const KmResidentArgs& a, hipStream_t stream)
{
    switch (a.D) {
    case 1: return launch_d<1>(a, stream);
    case 2: return launch_d<2>(a, stream);
    case 3: return launch_d<3>(a, stream);
    case 6: return launch_d<6>(a, stream);
    default: return false;
    }
}

}  // namespace mlhip
