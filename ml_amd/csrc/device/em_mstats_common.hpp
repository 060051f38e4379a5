// Shared pieces of the M-step statistics kernels (em_mstats.hip, em_mstats_wide.hip).
#pragma once
#include "device.hpp"

namespace mlhip {
namespace mstats {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TS = 64;            // samples per LDS tile
constexpr int XS = kRegDim + 3;   // LDS row stride of the sample tile in doubles: d+1 coordinates + zero slot, odd (35)
constexpr int XS_MID = kMidDim + 3;   // the same for 32 < d <= 64 (67)
constexpr int XS_BIG = kMaxDim + 3;   // and for 64 < d <= 128 (131)
template <int DM> constexpr int tile_stride() { return DM <= kRegDim ? XS : (DM <= kMidDim ? XS_MID : XS_BIG); }

/// Column `col` of the packed lower triangle of xt xt^T -> its (row a, column b) pair; padding columns map to the
/// zero slot `da` of the LDS row.
__device__ __forceinline__ void feature_pair(int col, int F, int da, int& a, int& b)
{
    if (col >= F) { a = b = da; return; }
    int r = (int)((__builtin_sqrt(8.0 * col + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= col) ++r;
    while (r * (r + 1) / 2 > col) --r;
    a = r;
    b = col - r * (r + 1) / 2;
}

/// Decomposition of the statistics GEMM over workgroups / waves.
struct Plan {
    bool wide;          // em_mstats_wide.hip (512 threads, waves split the column blocks) or em_mstats.hip (256 threads)
    bool small;         // em_mstats_small.hip (d <= 9: every wave on its own tile stream, all column blocks)
    int RBW, CBW;       // per-wave register blocking (16-row blocks x 16-column blocks)
    int RB, CB;         // total 16-blocks
    int n_rbg, n_cbg;   // grid.y decomposition
    int grid_x;
    int wg_per_cu;      // wide: 1, or 2 for the few-component shapes (see make_plan)
    int KP, FP;         // padded extents of one partial block
};
Plan make_plan(int d, int K, int num_cus);

int launch_wide(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream);
int launch_small(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream);
#ifdef MLHIP_EXPERIMENTS
int launch_narrow(const MstatsArgs& a, const Plan& p, int grid_x, hipStream_t stream);   // experiments/em_mstats_narrow.hip
#endif

}  // namespace mstats
}  // namespace mlhip
