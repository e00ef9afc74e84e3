// Device-side building block shared by spmv.hip / spgemm.hip / scan.hip:
// exclusive scan of one uint64 per thread across a workgroup (wave shuffles +
// one LDS hop).  blockDim.x must be a multiple of 64, at most 1024.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sprs_hip {

__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *wave_tot /*LDS, >= 16*/, uint64_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {
        const uint64_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// Workgroup barrier that orders LDS accesses only.  __syncthreads() carries a workgroup-scope release
// fence, which on gfx950 is an s_waitcnt vmcnt(0): the wave first waits for every global load AND store
// it has in flight.  Where the threads of a workgroup only hand LDS data to each other, that turns each
// barrier into a drain of unrelated streaming stores.
#ifndef SPRS_LDS_BARRIER   // (the CPU kernel emulator under tests/emu supplies its own two)
#define SPRS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SPRS_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
// the LDS operations this wave has issued so far are complete (no wait for global loads / stores in flight)
#define SPRS_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// inside a loop that polls an LDS word another wave of the workgroup will write
#define SPRS_POLL_PAUSE() __builtin_amdgcn_s_sleep(1)
#endif
__device__ __forceinline__ void lds_barrier() { SPRS_LDS_BARRIER(); }

// Hand-over of LDS data between the lanes of ONE wave.  __builtin_amdgcn_wave_barrier() alone only stops the compiler from
// moving code across it as a scheduling matter; it is not a memory fence, so LLVM could still forward or cache LDS values
// across it.  The wavefront-scope release / acquire pair makes the ordering part of the program (no instruction is emitted
// for it: the LDS operations of a wave execute in order anyway).
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive MAX scan over the 64 lanes (same DPP ladder as the sum; identity 0, operands are small non-negative numbers)
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v) {
    // UNSIGNED max (v_max_u32): a mark of 2^31 or more — a gap of that many rows inside one SpMM tile — stays the largest
    uint32_t x = v;
#define SPRS_MAX_STEP(CTRL, MASK, BOUND)                                                                  \
    {                                                                                                     \
        const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, MASK, 0xf, BOUND);     \
        x = o_ > x ? o_ : x;                                                                              \
    }
    SPRS_MAX_STEP(0x111, 0xf, true)      // row_shr:1 (zeros shifted in)
    SPRS_MAX_STEP(0x112, 0xf, true)      // row_shr:2
    SPRS_MAX_STEP(0x114, 0xf, true)      // row_shr:4
    SPRS_MAX_STEP(0x118, 0xf, true)      // row_shr:8
    SPRS_MAX_STEP(0x142, 0xa, false)     // row_bcast:15 into rows 1 and 3
    SPRS_MAX_STEP(0x143, 0xc, false)     // row_bcast:31 into rows 2 and 3
#undef SPRS_MAX_STEP
    return x;
}

// the same scans with LDS-only barriers
__device__ __forceinline__ uint64_t block_excl_scan_u64_lds(uint64_t v, uint64_t *wave_tot, uint64_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    uint64_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {
        const uint64_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    lds_barrier();
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ uint32_t block_excl_scan_u32_lds(uint32_t v, uint32_t *wave_tot, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    lds_barrier();
    *total = tot;
    return base + inc - v;
}

// 32-bit twin (half the shuffles) for sums known to stay below 2^32
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *wave_tot /*LDS, >= 16*/, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

}  // namespace sprs_hip
