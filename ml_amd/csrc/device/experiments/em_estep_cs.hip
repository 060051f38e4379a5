// EXPERIMENT (make EXPERIMENTS=1, MLHIP_ESTEP_CS=1) -- measured SLOWER than em_estep_mfma4.hip and not in the default library:
// N = 2.5M, d = 32, K = 64: 3.44 ms against 3.18 ms (rocprofv3; profiles/r03_estep_cs_*). Where the time goes (MLHIP_CS_DIAG
// timing variants, profiles/r03_estep_cs_diag.txt): re-loading the W blocks of the next component pair 11 %, the tile barrier
// 6.5 %, the epilogue 6.5 %; with all three removed the kernel is 5 % faster than em_estep_mfma4 -- the squares of y (which
// every form of this E-step needs) and the clock the chip holds under a denser matrix stream set the floor, not the operand
// feeding. tools/microbench_issue prices the instruction kinds next to the matrix stream (profiles/r03_microbench_issue.txt).
//
// Component-stationary E-step of Gaussian-mixture EM on the gfx950 fp64 matrix cores (dimensions 24..32, FOLD form, lw only)
// -- replaces EM::expectation_step (reference ML/EM.cpp:190-219) and its xAx_symmetric calls (ML/LinearAlgebra.cpp:8-31).
//
// Same arithmetic as em_estep_mfma4.hip in its FOLD / !LSE form -- y = W_k (x - s) - W_k (mu_k - s) with the second term as the
// accumulator initialiser, q = |y|^2, lw = coef_k - q/2, the 4x4 blocks of W on or below the diagonal on
// v_mfma_f64_4x4x4_4b_f64 in column-quad-major order -- but with the roles of the two matrix operands exchanged:
//
//   em_estep_mfma4:  a wave keeps 64 SAMPLES in registers (B operands) and walks the K components; the 36 blocks of W_k are
//                    A operands read from LDS, one read per 4 matrix instructions, one workgroup barrier per component.
//   here:            a wave keeps the W blocks of NC = 2 COMPONENTS in registers (A operands, 2 x 36 doubles per lane) and
//                    walks the samples of a 256-sample tile that the workgroup has staged in LDS as x - s; one B-operand read
//                    (ds_read_b64, conflict-free) feeds (Q - C) row quads x 2 components = 9 matrix instructions on average.
//                    No barrier inside a tile, no record staging; the records are read from global memory (L2) once per
//                    (wave, component pair, tile).
//
// Work split: a tile (256 samples = 4 groups of 64) x ceil(K/2) component pairs = 4 ceil(K/2) units, dealt to the 8 waves of
// the workgroup in contiguous runs (a wave changes its pair every 4 units). The next tile is fetched global -> registers -> LDS
// in 4-row chunks spread over the units (double-buffered tile, one barrier per tile).
//
// The log-sum-exp is not formed here (the K components of a sample are spread over the waves): the self-normalising
// statistics kernel does it (em_mstats_wide.hip), exactly as after em_estep_mfma4's !LSE form.
#include <cstdlib>
#include <type_traits>

#include "../device.hpp"

namespace mlhip {
namespace {

constexpr int kTile = kSampleTile;       // samples per tile
constexpr int kTilePitch = kTile + 16;   // doubles per tile row: rows g and g+1 of a B-operand read fall into disjoint banks
constexpr int kWaves = 8;
constexpr int kGroups = kTile / 64;      // 64-sample groups per tile

template <int D> struct Blocks {
    static constexpr int Q = D / 4;
    static constexpr int NB = Q * (Q + 1) / 2;
    static constexpr int PS = NB * 16 + 2 * D + 1;   // layout.hpp estep_mfma4_param_stride
};

/// See em_estep_mfma4.hip: lane group g = lane>>4 ends with the sum over groups of v[g].
__device__ __forceinline__ double reduce_scatter_groups(double v0, double v1, double v2, double v3)
{
    auto swap16 = [](double& a, double& b) {
        const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    };
    auto swap32 = [](double& a, double& b) {
        const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    };
    swap16(v0, v1);
    swap16(v2, v3);
    double t01 = v0 + v1, t23 = v2 + v3;
    swap32(t01, t23);
    return t01 + t23;
}

using double2_t = double __attribute__((ext_vector_type(2)));

// (An earlier version kept the accumulator initialisers packed in one register pair per component and unpacked them with DPP
// row_newbcast moves -- 64 more vector instructions per unit, each ~12 cycles when it sits between matrix instructions: 3.72 ms.)

/// q (+)= a^2, pinned where it is written (volatile): left to the compiler, the squares of one of the two components sink to the
/// end of the unit and its accumulators stay alive (64 registers). The operand was written by a matrix instruction at least four
/// matrix instructions (64 cycles) earlier -- see the placement in the kernel -- so no wait states are needed here.
template <bool FIRST> __device__ __forceinline__ void square_add(double& q, double a)
{
    if constexpr (FIRST) asm volatile("v_mul_f64 %0, %1, %1" : "=v"(q) : "v"(a));
    else asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(q) : "v"(a));
}

/// w = *(base + voff + OFF bytes), into the register w lives in (see reload_W_column in the kernel).
template <int OFF> __device__ __forceinline__ void load_in_place(double& w, int voff, const double* base)
{
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "+v"(w) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}

template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

/// CPU = tile chunks (4 rows x 256 samples) a thread carries in registers across one unit.
/// DIAG (timing diagnostics only, results WRONG): bit 0 = no tile barrier, bit 1 = records never reloaded, bit 2 = no
/// epilogue (no reduction, no stores).
template <int D, int CPU, int DIAG = 0>
__global__ __launch_bounds__(64 * kWaves, 2) void em_estep_cs_kernel(
    const double* __restrict__ xt, size_t ldx, uint32_t n_tiles, const double* __restrict__ params, int K,
    const double* __restrict__ shift, double* __restrict__ lw_out, size_t ldr, int pairs_per_wave)
{
    using B = Blocks<D>;
    constexpr int Q = B::Q, NB = B::NB, PS = B::PS;
    constexpr int NC = 2;                      // components per pass
    constexpr int NCH = D / 4;                 // chunks per tile
    constexpr int TILE = D * kTilePitch;       // doubles per tile buffer
    constexpr int ZW = 4, PD = 3;              // B-operand window / read-ahead distance (steps)
    extern __shared__ __attribute__((aligned(16))) double tile_dyn[];   // [2][D][kTilePitch]: x - shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int aoff = g * 4 + (lane & 3);       // A operand: entry [k = lane>>4][i = lane&3] of a 16-double block
    // tile staging role of this thread: row 4c + crow, sample pair ccol of chunk c
    const int crow = tid >> 7, ccol = (tid & 127) * 2;

    auto load_chunk = [&](uint32_t tile, int c) -> double2_t {
        const double* src = xt + (size_t)(4 * c + crow) * ldx + (size_t)tile * kTile + ccol;
        return *reinterpret_cast<const double2_t*>(src);
    };
    auto store_chunk = [&](int buf, int c, double2_t v, double sh) {
        v.x -= sh;
        v.y -= sh;
        *reinterpret_cast<double2_t*>(&tile_dyn[buf * TILE + (4 * c + crow) * kTilePitch + ccol]) = v;
    };
    // The records of a component pair: W blocks in A-operand layout, the accumulator initialisers -W (mu - s) (one register pair
    // per row quad) and coef. Every load of them is IN PLACE and pinned where it
    // is written (volatile asm, "+v": the destination is the very register the value lives in); left to the compiler, the
    // blocks of the next pair get registers of their own (+56 .. 90 registers), and its wait-count bookkeeping -- which
    // cannot tell which unit issued a load -- makes every unit wait for everything outstanding (the chunk just requested, the
    // previous unit's stores). The compiler does not see these loads: wait_for_records() follows before anything reads them.
    double W[NC][NB], init[NC][Q], coef[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int t = 0; t < NB; ++t) W[c][t] = 0.0;
#pragma unroll
        for (int R = 0; R < Q; ++R) init[c][R] = 0.0;
        coef[c] = 0.0;
    }
    const int aoff8 = aoff * 8, aoff8_hi = aoff * 8 + 4096;
    const int voff_init = (NB * 16 + D + g) * 8, voff_coef = (NB * 16 + 2 * D) * 8;
    auto reload_W_column = [&](int pair, auto C_) {
        constexpr int C = C_;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double* rec = params + (size_t)min(NC * pair + c, K - 1) * PS;
            static_for<Q - C>([&](auto r_) {
                constexpr int t = C * Q - C * (C - 1) / 2 + r_;
                constexpr bool hi = t * 128 >= 4096;
                load_in_place<t * 128 - (hi ? 4096 : 0)>(W[c][t], hi ? aoff8_hi : aoff8, rec);
            });
        }
    };
    auto reload_vectors = [&](int pair) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double* rec = params + (size_t)min(NC * pair + c, K - 1) * PS;
            static_for<Q>([&](auto R) { load_in_place<R * 32>(init[c][R], voff_init, rec); });
            load_in_place<0>(coef[c], voff_coef, rec);
        }
    };
    auto wait_for_records = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) store_chunk(0, c, load_chunk(tile, c), shift[4 * c + crow]);
    }
    // this wave's component pairs [p_begin, p_begin + pairs_per_wave) of every tile; a pair takes 4 units (64-sample groups)
    const int p_begin = wave * pairs_per_wave;
    static_for<Q>([&](auto C) { reload_W_column(p_begin, C); });
    reload_vectors(p_begin);
    wait_for_records();
    int buf = 0;

    // One unit: 4 sample blocks x Q column quads x (Q - C) row quads x NC components on the matrix cores. RELOAD: the wave
    // turns to another component pair after this unit -- its W blocks are requested column by column during the LAST sample
    // block, each column right after its last use, so that the records arrive under the remaining matrix work and their
    // loads are older than the stores of the epilogue (loads and stores complete in issue order).
    auto unit = [&](auto reload_, int pair, int next_pair, int sg, int j, bool has_next, uint32_t next) {
        constexpr bool RELOAD = reload_ && !(DIAG & 2);
        // next tile: chunks [j CPU, (j+1) CPU) travel through registers during the first three sample blocks
        double2_t pre[CPU];
        double presh[CPU];
#pragma unroll
        for (int i = 0; i < CPU; ++i) {
            const int c = j * CPU + i;
            if (has_next && c < NCH) {
                pre[i] = load_chunk(next, c);
                presh[i] = shift[4 * c + crow];
            }
        }
        // B operands of this unit: z(sbl, C) = tile[4C + g][64 sg + 16 sbl + s]
        const double* __restrict__ zt = tile_dyn + buf * TILE + g * kTilePitch + 64 * sg + s;
        auto zread = [&](int n) { return zt[(n % Q) * 4 * kTilePitch + (n / Q) * 16]; };
        double zw[ZW], acc[NC][Q], qs[4][NC];
#pragma unroll
        for (int n = 0; n < PD; ++n) zw[n % ZW] = zread(n);
        static_for<4>([&](auto sbl_) {
            constexpr int sbl = sbl_;
            if constexpr (sbl == 3) {
#pragma unroll
                for (int i = 0; i < CPU; ++i) {
                    const int c = j * CPU + i;
                    if (has_next && c < NCH) store_chunk(buf ^ 1, c, pre[i], presh[i]);
                }
            }
            static_for<Q>([&](auto C_) {
                constexpr int C = C_;
                constexpr int n = sbl * Q + C;
                const double z = zw[n % ZW];
                if (n + PD < 4 * Q) zw[(n + PD) % ZW] = zread(n + PD);
#pragma unroll
                for (int R = C; R < Q; ++R) {
                    const int t = C * Q - C * (C - 1) / 2 + (R - C);   // column-quad-major block index
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        acc[c][R] = __builtin_amdgcn_mfma_f64_4x4x4f64(W[c][t], z, C == 0 ? init[c][R] : acc[c][R], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);   // pin the block order
                }
                if constexpr (RELOAD && sbl == 3) {
                    reload_W_column(next_pair, C_);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // |y|^2 of this sample block, all 2 Q squares in ONE cluster behind the block's matrix instructions: a vector
            // instruction BETWEEN two matrix instructions costs the SIMD ~12 cycles, in a cluster ~5 (tools/microbench_issue)
            static_for<Q>([&](auto R_) {
                constexpr int R = R_;
#pragma unroll
                for (int c = 0; c < NC; ++c) square_add<R == 0>(qs[sbl][c], acc[c][R]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (DIAG & 4) {
            if (qs[0][0] + qs[1][0] + qs[2][0] + qs[3][0] + qs[0][1] + qs[1][1] + qs[2][1] + qs[3][1] == 1.2345e-300) lw_out[lane] = 0.0;
            return;
        }
        double lw[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double q = reduce_scatter_groups(qs[0][c], qs[1][c], qs[2][c], qs[3][c]);   // lane (g, s): sample 16g + s
            lw[c] = __builtin_fma(-0.5, q, coef[c]);
        }
        if constexpr (RELOAD) {
            asm volatile("" : "+v"(lw[0]), "+v"(lw[1]));   // after the reduction (which still reads coef)
            reload_vectors(next_pair);
            // every load of the new records has landed before the stores are issued: at the top of the next unit nothing the
            // matrix instructions need is in flight any more (a wait there would also wait for these stores to reach memory)
            wait_for_records();
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = NC * pair + c;
            if (k < K) lw_out[(size_t)k * ldr + (size_t)tile * kTile + 64 * sg + lane] = lw[c];
        }
    };

    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        if constexpr (!(DIAG & 1)) __syncthreads();   // tile `tile` complete in buffer buf; every wave is done reading the other buffer
        const uint32_t next = tile + gridDim.x;
        const bool has_next = next < n_tiles;
        int j = 0;   // unit slot of the tile: slot j carries chunks [j CPU, (j+1) CPU) of the next tile
#pragma unroll 1
        for (int pi = 0; pi < pairs_per_wave; ++pi) {
            const int pair = p_begin + pi;
            const int next_pair = pi + 1 < pairs_per_wave ? pair + 1 : p_begin;
            // ONE path through this loop body, and only its last unit touches W (in place): no copies of the W registers at
            // the loop's joins (two alternative unit bodies per slot made the compiler keep two register sets)
#pragma unroll 1
            for (int sg = 0; sg < kGroups - 1; ++sg, ++j) unit(std::false_type{}, pair, next_pair, sg, j, has_next, next);
            unit(std::true_type{}, pair, next_pair, kGroups - 1, j, has_next, next);
            ++j;
        }
    }
}

template <int D, int CPU, int DIAG = 0>
int launch_cpu(const EstepArgs& a, int num_cus, int ppw, hipStream_t stream)
{
    constexpr size_t smem = sizeof(double) * 2 * D * kTilePitch;
    static bool attr_set = false;   // > 64 KB of dynamic LDS needs the opt-in once per process
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&em_estep_cs_kernel<D, CPU, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return -1;
        attr_set = true;
    }
    const uint32_t n_tiles = padded_samples(a.n) / kTile;
    uint32_t grid = n_tiles < (uint32_t)num_cus ? n_tiles : (uint32_t)num_cus;
    hipLaunchKernelGGL((em_estep_cs_kernel<D, CPU, DIAG>), dim3(grid), dim3(64 * kWaves), smem, stream, a.xt, a.ldx, n_tiles, a.params,
                       a.K, a.shift, a.lw, a.ldr, ppw);
    return (int)grid;
}

template <int D>
int launch_d(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    const int ppw = (a.K + 1) / 2 / kWaves;           // component pairs per wave
    const int cpu = (D / 4 + 4 * ppw - 1) / (4 * ppw);   // chunks per unit slot
    if constexpr (D == 32) {
        static const int diag = [] { const char* e = std::getenv("MLHIP_CS_DIAG"); return e ? std::atoi(e) : 0; }();
        if (cpu == 1 && diag == 1) return launch_cpu<D, 1, 1>(a, num_cus, ppw, stream);
        if (cpu == 1 && diag == 2) return launch_cpu<D, 1, 2>(a, num_cus, ppw, stream);
        if (cpu == 1 && diag == 3) return launch_cpu<D, 1, 3>(a, num_cus, ppw, stream);
        if (cpu == 1 && diag == 4) return launch_cpu<D, 1, 4>(a, num_cus, ppw, stream);
        if (cpu == 1 && diag == 7) return launch_cpu<D, 1, 7>(a, num_cus, ppw, stream);
    }
    if (cpu == 1) return launch_cpu<D, 1>(a, num_cus, ppw, stream);
    if (cpu == 2) return launch_cpu<D, 2>(a, num_cus, ppw, stream);
    return -1;
}

}  // namespace

/// Whether the component-stationary kernel serves (D, K): the component pairs of a tile are dealt to the 8 waves whole and
/// evenly (ceil(K/2) a multiple of 8), and a wave's 4 * pairs unit slots must carry the D/4 chunks of the next tile at no
/// more than two per slot.
bool em_estep_cs_supported(int D, int K)
{
    if (D != 24 && D != 28 && D != 32) return false;
    const int pairs = (K + 1) / 2;
    // D = 32 with two chunks per unit slot (K = 16) spills registers next to the in-place record loads, which the compiler does
    // not know about: a spilled block is saved before its load has landed (results differed from em_estep_mfma4) -- refused
    if (D == 32 && pairs / kWaves < 2) return false;
    return pairs >= kWaves && pairs % kWaves == 0;
}

int launch_em_estep_cs(const EstepArgs& a, int num_cus, hipStream_t stream)
{
    if (!a.fold || a.with_lse || !em_estep_cs_supported(a.D, a.K)) return -1;
    switch (a.D) {
    case 24: return launch_d<24>(a, num_cus, stream);
    case 28: return launch_d<28>(a, num_cus, stream);
    case 32: return launch_d<32>(a, num_cus, stream);
    default: return -1;
    }
}

}  // namespace mlhip
