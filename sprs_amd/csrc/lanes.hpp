// Lane exchanges of one wavefront WITHOUT the LDS crossbar (gfx950): DPP modifiers inside a row of 16 lanes, v_permlane16_swap /
// v_permlane32_swap between rows — VALU instructions that fold into their consumer (v_min_u32_dpp ...), where __shfl_xor is a
// ds_bpermute_b32 plus an address and a wait per exchange.  The micro-row kernel of spgemm.hip sorts a lane group with them
// (a bitonic network: 10 / 15 / 21 exchanges for 16 / 32 / 64 lanes).  The CPU emulator (tests/emu) models the DPP controls and
// the two swaps and runs these sequences as written; scripts/probes/lane_ops.hip checks every function here against __shfl on
// the hardware.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sprs_hip {

// the value of lane (own ^ J), J a power of two below 64
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "a single bit");
    if constexpr (J == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);       // quad_perm:[1,0,3,2]
    } else if constexpr (J == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);       // quad_perm:[2,3,0,1]
    } else if constexpr (J == 4) {
        // no single modifier: banks 0 and 2 of a row take lane + 4 (row_shl:4), banks 1 and 3 lane - 4 (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xA, false);
    } else if constexpr (J == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);      // row_ror:8
    } else if constexpr (J == 16) {
        // the instruction swaps the odd rows of its first operand with the even rows of its second: of two copies of v the
        // first then holds rows {0, 0, 2, 2}, the second rows {1, 1, 3, 3}
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16u) ? r[0] : r[1];
    } else {
        // ... the upper half of the first with the lower half of the second: halves {lo, lo} and {hi, hi}
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32u) ? r[0] : r[1];
    }
}

template <int J>
__device__ __forceinline__ uint64_t lane_xor(uint64_t v) {
    return (uint64_t)lane_xor<J>((uint32_t)v) | ((uint64_t)lane_xor<J>((uint32_t)(v >> 32)) << 32);
}

// the value of the lane below (lane - 1); lane 0 of the wave gets 0
__device__ __forceinline__ uint32_t lane_below(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);          // wave_shr:1
}

// the value of the lane above (lane + 1); lane 63 gets 0
__device__ __forceinline__ uint32_t lane_above(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);          // wave_shl:1
}

// inclusive sum over the lanes of a GROUP of G = 16 / 32 / 64 consecutive lanes (groups aligned to G)
template <int G>
__device__ __forceinline__ uint32_t group_incl_scan_u32(uint32_t v) {
    static_assert(G == 16 || G == 32 || G == 64, "a row, two rows or the wave");
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);     // row_shr:1 (zeros shifted in)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);     // row_shr:8
    if constexpr (G >= 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    if constexpr (G >= 64) x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// ascending sort of one key per lane inside every group of G lanes (keys of a group distinct, or equal keys interchangeable):
// the bitonic network, every exchange a lane_xor
template <int G, typename KEY>
__device__ __forceinline__ KEY group_sort(KEY key) {
    const uint32_t gl = threadIdx.x & (uint32_t)(G - 1);
#define SPRS_SORT_STEP(K2, J)                                                                             \
    if constexpr (K2 <= G && J < K2) {                                                                    \
        const KEY p_ = lane_xor<J>(key);                                                                  \
        const bool take_min_ = ((gl & (uint32_t)J) == 0) == ((gl & (uint32_t)K2) == 0);                   \
        const KEY lo_ = p_ < key ? p_ : key, hi_ = p_ < key ? key : p_;                                   \
        key = take_min_ ? lo_ : hi_;                                                                      \
    }
#define SPRS_SORT_STAGE(K2)                                                                               \
    SPRS_SORT_STEP(K2, 32) SPRS_SORT_STEP(K2, 16) SPRS_SORT_STEP(K2, 8) SPRS_SORT_STEP(K2, 4) SPRS_SORT_STEP(K2, 2) SPRS_SORT_STEP(K2, 1)
    SPRS_SORT_STAGE(2) SPRS_SORT_STAGE(4) SPRS_SORT_STAGE(8) SPRS_SORT_STAGE(16) SPRS_SORT_STAGE(32) SPRS_SORT_STAGE(64)
#undef SPRS_SORT_STAGE
#undef SPRS_SORT_STEP
    return key;
}

}  // namespace sprs_hip
