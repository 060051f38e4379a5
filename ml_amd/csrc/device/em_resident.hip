// The WHOLE loop of EM::fit (reference ML/EM.cpp:143-170: E-step, M-step, convergence test, until converged or maximum_steps) in
// ONE launch, for fits whose iteration is a few microseconds of arithmetic -- the reference's own benchmark sizes
// (Benchmarks/bm_EM.cpp: d = 2, K = 3, N = 100 ... 100 000). As three dependent launches per iteration (E+M kernel, reduction,
// closing kernel) such an iteration costs 18 - 21 us of which ~9 us is kernel time: the rest is dispatch, and a HIP graph or a
// last-workgroup tail does not remove it. Here the workgroups stay resident for the whole fit:
//
//   per iteration, every workgroup
//     A. runs the vector-unit E+M pass over its tiles (em_fused_valu_body.hpp: the same text as em_fused_valu_kernel, records from
//        an LDS copy instead of scalar registers) and PUBLISHES its partial block [K F statistics | log-likelihood sum] with
//        write-through (sc1) stores, then one agent-scope arrival add;
//     B. waits for the arrivals of all workgroups (one wave polls one word), reads ALL partial blocks back with sc1 loads and sums
//        them in the order of em_reduce_kernel (em_mstats.hip) -- every workgroup forms the same sums, bit for bit;
//     C. closes the iteration itself (em_close_body.hpp, one wave per component): new parameters, the next records -- straight into
//        its LDS copy -- and the reference's convergence test (ML/EM.cpp:161-168), all redundantly and identically in every
//        workgroup, so that ONE hand-off per iteration is all the workgroups exchange. Workgroup 0 also writes parameters, records
//        and the log-likelihood history where the host expects them (the ring of runtime/em_loop.cpp).
//
// The exchange follows the measured hand-off form of the gfx950 guide (one lane per storing workgroup signals for all its stores
// behind every storing wave's s_waitcnt vmcnt(0) and a workgroup barrier; the consumer polls with an sc1 load; EVERY load of the
// handed-off bytes is an sc1 load to registers behind that poll and a workgroup barrier; at most one workgroup per CU; hipMalloc
// memory): the per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores, so nothing
// here relies on a plain load of another workgroup's data. Two exchange buffers alternate: a workgroup can be at most one hand-off
// ahead of the slowest one. Every spin is bounded: a workgroup that waits too long sets the status word and leaves, the others
// follow; the host then runs the ordinary loop from the caller's starting values.
//
// Bit-identical to the three-launch loop by construction: same pass, same partial blocks (the grid is the one em_fused_valu_kernel
// would get), same summation order, same closing arithmetic (tests/test_gpu_resident.py).
#include "em_close_body.hpp"
#include "em_fused_valu_body.hpp"
#include "parts.hpp"

namespace mlhip {
namespace mstats {
namespace {

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(3))) const double lds_cdouble;


constexpr unsigned kSpinLimit = 1u << 22;      // polls of ~1 us: a few seconds, then the workgroup gives up (status 3)

/// MLHIP_RESIDENT_PROFILE=1 (tools/resident_phases.py): thread 0 of workgroup 0 stamps the constant 100 MHz clock at the phase
/// boundaries of every iteration (kResidentStamps slots each: 0 - 7 the phases of the loop below, 8 - 11 inside the pass, 12 - 17
/// inside the closing arithmetic); a null pointer otherwise -- one scalar branch per boundary.
__device__ __forceinline__ void stamp(unsigned long long* prof, uint32_t i, int slot, uint32_t g, int tid)
{
    if (prof && g == 0 && tid == 0) prof[(size_t)i * kResidentStamps + slot] = wall_clock64();
}
struct StampProbe {
    unsigned long long* prof; uint32_t i, g; int tid;
    __device__ __forceinline__ void operator()(int slot) const { stamp(prof, i, slot, g, tid); }
};

/// One exchanged value: 16 bytes = two 8-byte granules {half of the double, tag}, written by ONE write-through (sc1) store and read
/// by ONE sc1 load; a value is there when BOTH its granules carry the iteration's tag (each aligned 8-byte half arrives whole).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;                                        // cache-policy bit of the raw buffer intrinsics: sc1
__device__ __forceinline__ u32x4 tagged(double v, unsigned tag)
{
    return u32x4{(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
}

/// Every workgroup's partial block of iteration `epoch - 1` -> vals[b * XS + e] in LDS. A thread takes the values p = tid, tid + 256,
/// ... of the G XS exchanged ones (starting at byte `base` of the exchange buffer); ALL its loads are issued before the first is
/// looked at, and they are read again until every granule carries this iteration's tag: waiting for the other workgroups and
/// gathering their sums is ONE round trip (~0.9 us to the memory side and back: write-through stores leave no copy in any L2)
/// when they are on time. (Two reads in flight a quarter of a microsecond apart, so that a value landing just behind the first is
/// caught by the second, were measured SLOWER at 40 workgroups -- 3.9 against 2.3 us: the all-to-all read is G^2 blocks of
/// uncached requests, and doubling them costs more than the saved wait.)
/// Returns false when the wait gave up (bounded spin, or another workgroup raised the flag).
template <int XS>
__device__ __forceinline__ bool gather_blocks(__amdgpu_buffer_rsrc_t xch, unsigned base, uint32_t G, unsigned epoch, int tid, double* vals,
                                              gu32* give_up)
{
    constexpr int MAXP = (kResidentMaxGrid * XS + 255) / 256;      // values per thread at the largest grid
    const int total = (int)G * XS;
    u32x4 v[MAXP];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int p = tid + 256 * j;
            if (p < total) v[j] = __builtin_amdgcn_raw_buffer_load_b128(xch, base + 16u * (unsigned)p, 0, kAuxSc1);
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
            if (tid + 256 * j < total) ok = ok && v[j].y == epoch && v[j].w == epoch;
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))
            return false;
    }
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int p = tid + 256 * j;
        if (p < total) vals[p] = __hiloint2double((int)v[j].z, (int)v[j].x);
    }
    return true;
}

template <int D, int K>
__global__ __launch_bounds__(256) void em_resident_valu_kernel(ResidentArgs a)
{
#pragma clang fp contract(off)     // the convergence test is the host's arithmetic, statement by statement (em_loop.cpp ConvergenceTest)
    using S = ValuShape<D, K>;
    constexpr int PS = S::PS, F = S::F, VP = S::VP, TOT = K * F;
    constexpr int XS = TOT + 1;                                    // values per exchanged block: [K F sums | log-likelihood sum]
    constexpr int CS = (int)closing::scratch_doubles(D);
    __shared__ double fold[4][VP];
    __shared__ double red[4];
    __shared__ double vals[kResidentMaxGrid * XS];                 // every workgroup's partial block of this iteration
    __shared__ double stats[TOT + 1];
    __shared__ __attribute__((aligned(16))) double recs[K * PS];
    __shared__ double o_mixing[K], o_means[K * D], o_covs[K * D * D], o_info[1 + 2 * K];
    __shared__ double scratch[4][CS];
    __shared__ int s_gave_up;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x, g = blockIdx.x;                  // one resident workgroup per partial block
    const uint32_t n_tiles = (a.n + TS - 1) / TS;
    gu32* give_up = (gu32*)a.sync;
    // the exchange buffer as a raw buffer (byte offsets, bounds-checked by the hardware): 16-byte sc1 loads / stores with the compiler's
    // own wait counters
    const __amdgpu_buffer_rsrc_t xch = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, (int)(2u * G * XS * 16u), 0x00020000);

    for (int e = tid; e < K * PS; e += 256) recs[e] = a.records[0][e];
    if (tid == 0) s_gave_up = 0;
    double xn[D];
    const uint32_t first_tile = g * 4 + wave < n_tiles ? g * 4 + wave : 0;
    valu_load_tile<D>(a.xt, a.ldx, first_tile, lane, xn);
    __syncthreads();

    double old_ll = -__builtin_inf();
    uint32_t status = 1, steps = 0, converged = 0;
    for (uint32_t i = 0; i < a.max_steps; ++i) {
        const int out = (int)((i + 1) % 3);
        const unsigned epoch = i + 1;                              // tag of this iteration's granules (never 0: the buffers start zeroed)
        const unsigned xb = (i & 1u) * G * XS * 16u;               // byte offset of this iteration's buffer
        stamp(a.profile, i, 0, g, tid);
        // ---- A. E-step + statistics over this workgroup's tiles; the partial block published as tagged granules, write-through
        {
            double acc[VP];
#pragma unroll
            for (int e = 0; e < VP; ++e) acc[e] = 0.0;
            double ll_acc = 0.0;
            valu_tiles<D, K>(a.xt, a.ldx, a.n, a.shift, (lds_cdouble*)recs, (double*)nullptr, g, G, wave, lane, xn, acc, ll_acc,
                             StampProbe{a.profile, i, g, tid});
            stamp(a.profile, i, 11, g, tid);
            valu_fold<VP>(acc, ll_acc, wave, lane, fold, red);
            __syncthreads();
            stamp(a.profile, i, 1, g, tid);
            for (int e = tid; e < XS; e += 256) {
                const double v = e < TOT ? ((fold[0][e] + fold[1][e]) + fold[2][e]) + fold[3][e] : ((red[0] + red[1]) + red[2]) + red[3];
                __builtin_amdgcn_raw_buffer_store_b128(tagged(v, epoch), xch, xb + 16u * (g * XS + (unsigned)e), 0, kAuxSc1);
            }
        }
        stamp(a.profile, i, 2, g, tid);
        valu_load_tile<D>(a.xt, a.ldx, first_tile, lane, xn);      // the next iteration's first tile: in flight behind the exchange and the closing
        // ---- B. every workgroup's partial block, as soon as it is there, summed in em_reduce_kernel's order (em_mstats.hip)
        if (!gather_blocks<XS>(xch, xb, G, epoch, tid, vals, give_up)) s_gave_up = 1;
        __syncthreads();
        if (s_gave_up) {                                           // (workgroup-uniform) somebody never published: tell the others, leave
            if (tid == 0) __hip_atomic_store(give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            status = 3;
            break;
        }
        stamp(a.profile, i, 4, g, tid);
        if (tid < TOT) {
            // sum e: 32 slice sums s_q = 0 + v[q][e] + v[q + 32][e] + ... (blocks ascending), added in ascending q. A block that does
            // not exist adds +0.0, which leaves a sum -- never -0.0, it starts at +0.0 -- as it is. Every read is issued before the
            // first addition (reads of blocks beyond the grid go to block 0 and are replaced by +0.0).
            static_assert(kResidentMaxGrid == 2 * kResidentSlices, "two blocks per slice at most");
            double lo[kResidentSlices], up[kResidentSlices];
#pragma unroll
            for (int q = 0; q < kResidentSlices; ++q) lo[q] = vals[((uint32_t)q < G ? q : 0) * XS + tid];
            if (G > (uint32_t)kResidentSlices) {
#pragma unroll
                for (int q = 0; q < kResidentSlices; ++q) up[q] = vals[((uint32_t)(q + kResidentSlices) < G ? q + kResidentSlices : 0) * XS + tid];
            } else {
#pragma unroll
                for (int q = 0; q < kResidentSlices; ++q) up[q] = 0.0;
            }
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < kResidentSlices; ++q) {
                // (slices q >= G are empty: their +0.0 would change nothing -- t is never -0.0 -- so the chain of additions ends at G)
                if ((uint32_t)q < G) {
                    double sq = 0.0;
                    sq += lo[q];
                    sq += (uint32_t)(q + kResidentSlices) < G ? up[q] : 0.0;
                    t = q == 0 ? sq : t + sq;
                }
            }
            stats[tid] = t;
        }
        if (wave == 3) {
            // the log-likelihood sums: em_reduce_kernel's tree over 256 per-thread sums s_t = 0 + ll[t] + ll[t + 256] + ... (buf[t] +=
            // buf[t + off], off = 128 ... 1) by one wave: lane l holds t = l, l + 64, (l + 128, l + 192: zero, the grid is at most 64);
            // the steps below 64 are lane shifts (lanes >= off compute values nobody reads)
            static_assert(kResidentMaxGrid <= 128, "the tree below assumes at most 128 partial blocks");
            double v0 = (uint32_t)lane < G ? 0.0 + vals[lane * XS + TOT] : 0.0;
            double v1 = (uint32_t)lane + 64 < G ? 0.0 + vals[(lane + 64) * XS + TOT] : 0.0;
            v0 += 0.0; v1 += 0.0;
            v0 += v1;
            v0 += upper_half<true>(v0);
            v0 += upper_half<false>(v0);
            v0 += row_shift_left<8>(v0);
            v0 += row_shift_left<4>(v0);
            v0 += row_shift_left<2>(v0);
            v0 += row_shift_left<1>(v0);
            if (lane == 0) stats[TOT] = v0;
        }
        __syncthreads();
        stamp(a.profile, i, 5, g, tid);
        // ---- C. closing arithmetic, one wave per component (the four waves take turns): parameters, the next records, flags
        for (int k = wave; k < K; k += 4)
            closing::close_component<0, D, true>(stats, K, D, D, a.shift, a.n_global, a.refine_limit, o_mixing, o_means, o_covs, recs, PS,
                                           o_info, k, lane, scratch[wave], StampProbe{a.profile, i, g, tid});
        __syncthreads();
        stamp(a.profile, i, 6, g, tid);
        const double ll = o_info[0] / a.n_global - a.ll_offset;    // ML/EM.cpp:197-198, 211
        bool flagged = false;
#pragma unroll
        for (int k = 0; k < K; ++k) flagged = flagged || o_info[1 + k] != 0.0;
        // what the three launches leave behind for the host -- pack and records of ring slot (i + 1) % 3, the history -- written by
        // up to four workgroups, a part each (every workgroup holds all of it)
        if (g == 0) {
            for (int e = tid; e < 1 + 2 * K; e += 256) a.info[out][e] = o_info[e];
            if (tid == 0) a.history[i] = ll;
        }
        if (g == 1 % G) {
            for (int e = tid; e < K; e += 256) a.mixing[out][e] = o_mixing[e];
            for (int e = tid; e < K * D; e += 256) a.means[out][e] = o_means[e];
        }
        if (g == 2 % G)
            for (int e = tid; e < K * D * D; e += 256) a.covs[out][e] = o_covs[e];
        if (g == 3 % G)
            for (int e = tid; e < K * PS; e += 256) a.records[out][e] = recs[e];
        stamp(a.profile, i, 7, g, tid);
        if (flagged) { status = 2; steps = i; break; }             // a far, tight component: the host closes this iteration itself
        steps = i + 1;
        if (i > 0) {
            const double change = fabs(ll - old_ll);
            const double scale = fabs(old_ll) > fabs(ll) ? fabs(old_ll) : fabs(ll);
            if (change < a.atol + a.rtol * scale) { converged = 1; break; }
        }
        old_ll = ll;
    }
    if (g == 0 && tid == 0) {
        a.result[1] = steps;
        a.result[2] = converged;
        a.result[0] = status;
    }
}

template <int D, int K> bool launch_k(const ResidentArgs& a, int grid, hipStream_t stream)
{
    if constexpr (K >= 1) {
        if (a.K == K) {
            hipLaunchKernelGGL((em_resident_valu_kernel<D, K>), dim3(grid), dim3(256), 0, stream, a);
            return true;
        }
        return launch_k<D, K - 1>(a, grid, stream);
    }
    return false;
}

/// Largest K at dimension D the resident loop is built for: the shapes the vector-unit form takes at EVERY sample count
/// (K F <= 64 accumulators per lane, valu_form_applies in em_fused_small.hip).
/// ... and at most 16 components: the four waves of a workgroup close them in turns of ~2.5 us each, so beyond four turns the
/// serial closing costs more than the dispatches it saves (N = 16 384, d = 1, K = 21: 21.0 against 18.0 us per iteration with
/// three launches, profiles/r05_resident_phases.txt).
constexpr int resident_max_k(int D) { return 64 / ((D + 1) * (D + 2) / 2) < 16 ? 64 / ((D + 1) * (D + 2) / 2) : 16; }

}  // namespace

// ---- compiled in five parts by dimension (parts.hpp): part 1 .. 5 = d 1, 2, 3, 4, 6
bool MLHIP_PART_FN(launch_em_resident)(const ResidentArgs& a, hipStream_t stream)
{
    constexpr int D = MLHIP_PART <= 4 ? MLHIP_PART : 6;
    return a.d == D && launch_k<D, resident_max_k(D)>(a, a.vgrid, stream);   // one resident workgroup per partial block (<= one per CU)
}

#if MLHIP_PART == 1
bool launch_em_resident_part2(const ResidentArgs&, hipStream_t);
bool launch_em_resident_part3(const ResidentArgs&, hipStream_t);
bool launch_em_resident_part4(const ResidentArgs&, hipStream_t);
bool launch_em_resident_part5(const ResidentArgs&, hipStream_t);

/// Two alternating buffers of vgrid blocks of K F + 1 values, two 8-byte granules {tag, half} per value.
size_t em_resident_exchange_doubles(int d, int K, int vgrid) { return 2 * (size_t)vgrid * ((size_t)K * stats_count(d) + 1) * 2; }

bool em_resident_supported(int d, int K, int vgrid, int num_cus)
{
    if (padded_dim(d) != d || d > 6 || K < 1 || vgrid < 1 || vgrid > num_cus || vgrid > kResidentMaxGrid) return false;
    switch (d) {
    case 1: return K <= resident_max_k(1);
    case 2: return K <= resident_max_k(2);
    case 3: return K <= resident_max_k(3);
    case 4: return K <= resident_max_k(4);
    case 6: return K <= resident_max_k(6);
    default: return false;
    }
}

bool launch_em_resident(const ResidentArgs& a, hipStream_t stream)
{
    switch (a.d) {
    case 1: return launch_em_resident_part1(a, stream);
    case 2: return launch_em_resident_part2(a, stream);
    case 3: return launch_em_resident_part3(a, stream);
    case 4: return launch_em_resident_part4(a, stream);
    case 6: return launch_em_resident_part5(a, stream);
    default: return false;
    }
}
#endif

}  // namespace mstats
}  // namespace mlhip
