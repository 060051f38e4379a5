// Cross-lane moves of a double inside a 64-lane wavefront on the vector unit (DPP row shifts, v_permlane swaps): a few cycles each,
// where a ds_bpermute round trip through the LDS pipe is > 100. Shared by the vector-unit E+M pass (em_fused_valu_body.hpp), the
// closing arithmetic (em_close_body.hpp) and the device-resident loop (em_resident.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mlhip {

/// Lane l <- lane l + OFF of its 16-lane row (OFF = 8, 4, 2, 1; lanes whose source lies beyond the row get 0.0).
template <int OFF> __device__ __forceinline__ double row_shift_left(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + OFF, 0xf, 0xf, true);   // row_shl:OFF
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + OFF, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

/// Lanes 0-31 <- lanes 32-63 (BIT5) / the even 16-lane rows <- the odd ones: v_permlane32_swap / v_permlane16_swap of v with itself.
template <bool BIT5> __device__ __forceinline__ double upper_half(double v)
{
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (BIT5)
        return __hiloint2double((int)__builtin_amdgcn_permlane32_swap(hi, hi, false, false)[1], (int)__builtin_amdgcn_permlane32_swap(lo, lo, false, false)[1]);
    else
        return __hiloint2double((int)__builtin_amdgcn_permlane16_swap(hi, hi, false, false)[1], (int)__builtin_amdgcn_permlane16_swap(lo, lo, false, false)[1]);
}

/// Lane 0 <- max over the 64 lanes of a NON-NEGATIVE, non-NaN value (the other lanes: partial results). Exact, order-free.
__device__ __forceinline__ double wave_max_nonneg(double v)
{
    v = fmax(v, upper_half<true>(v));
    v = fmax(v, upper_half<false>(v));
    v = fmax(v, row_shift_left<8>(v));
    v = fmax(v, row_shift_left<4>(v));
    v = fmax(v, row_shift_left<2>(v));
    v = fmax(v, row_shift_left<1>(v));
    return v;
}

/// Diagnostic hook of the device functions of em_fused_valu_body.hpp and em_close_body.hpp: `probe(slot)` marks a point in the
/// instruction stream. The default does nothing and costs nothing; em_resident.hip passes one that stamps the clock
/// (MLHIP_RESIDENT_PROFILE=1).
struct NoProbe { __device__ __forceinline__ void operator()(int) const {} };

}  // namespace mlhip
