// Stable LSD radix sort of (key, value) pairs on the device — building block of
//   * the triplet -> CSR / CSC assembly (twin of TriMatIter::into_cs, sprs/src/sparse/triplet_iter.rs:127-224: sort by
//     (outer, inner), then fold neighbours; here the sort is STABLE, so duplicates are folded in triplet order), and
//   * the window-major order of the SpGEMM tasks (spgemm.hip).
// 8 bits per pass; every wave owns a contiguous chunk of the input and ranks its elements with eight ballots per 64
// elements (the lanes holding the same digit find each other without a loop); integer-only, deterministic, no atomics
// on the data path.  HBM-bound: 32 B per element and pass (16 in, 16 out).
#include "common.hpp"

#include <vector>

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr int RS_BLOCK = 256;
constexpr int RS_WAVES = RS_BLOCK / WAVE;
constexpr int RS_CHUNK = 4096;          // elements per wave
constexpr int RS_BINS = 256;

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// hist[d * nchunks + c] = number of elements of chunk c whose digit is d
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n, int shift, uint32_t mask,
                                                           uint64_t nchunks, uint64_t *__restrict__ hist) {
    __shared__ uint32_t h[RS_WAVES][RS_BINS];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const uint64_t c = (uint64_t)blockIdx.x * RS_WAVES + wave;
    for (uint32_t d = lane; d < (uint32_t)RS_BINS; d += WAVE) h[wave][d] = 0;
    wave_fence();
    if (c < nchunks) {
        const uint64_t lo = c * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
        for (uint64_t i = lo + lane; i < hi; i += WAVE) atomicAdd(&h[wave][(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    wave_fence();
    if (c < nchunks)
        for (uint32_t d = lane; d < (uint32_t)RS_BINS; d += WAVE) hist[(uint64_t)d * nchunks + c] = h[wave][d];
}

// offs = exclusive scan of hist (digit-major): where chunk c's first element of digit d goes
__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                                              uint64_t n, int shift, uint32_t mask, uint64_t nchunks,
                                                              const uint64_t *__restrict__ offs, uint64_t *__restrict__ keys_out,
                                                              uint64_t *__restrict__ vals_out) {
    __shared__ uint64_t next_s[RS_WAVES][RS_BINS];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint64_t c = (uint64_t)blockIdx.x * RS_WAVES + wave;
    if (c >= nchunks) return;                                   // wave-uniform; no workgroup barrier in this kernel
    uint64_t *next = next_s[wave];
    for (uint32_t d = lane; d < (uint32_t)RS_BINS; d += WAVE) next[d] = offs[(uint64_t)d * nchunks + c];
    wave_fence();
    const uint64_t lo = c * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
    for (uint64_t i0 = lo; i0 < hi; i0 += WAVE) {
        const uint64_t i = i0 + lane;
        const bool valid = i < hi;
        const uint64_t k = valid ? keys[i] : 0ull, v = valid ? vals[i] : 0ull;
        const uint32_t d = (uint32_t)(k >> shift) & mask;
        // lanes with my digit: intersect, over the 8 digit bits, the ballot of "bit set" or its complement
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below), cnt = (uint32_t)__popcll(same);
        uint64_t base = 0;
        if (valid) base = next[d];
        wave_fence();                                           // everyone has read its base ...
        if (valid && rank == 0) next[d] = base + cnt;           // ... before the first lane of each digit advances it
        wave_fence();
        if (valid) {
            keys_out[base + rank] = k;
            vals_out[base + rank] = v;
        }
    }
}

}  // namespace

// Sorts n (key, value) pairs in place, stably, by the key bits named in `fields` = {shift, nbits} pairs, least
// significant field first (bits outside the fields do not take part).
int32_t radix_sort_pairs(uint64_t *keys, uint64_t *vals, uint64_t n, const std::vector<std::pair<int, int>> &fields, hipStream_t stream) {
    if (n < 2) return SPRS_HIP_OK;
    std::vector<std::pair<int, uint32_t>> passes;              // shift, mask
    for (const auto &f : fields)
        for (int b = 0; b < f.second; b += 8) passes.push_back({f.first + b, (1u << (f.second - b < 8 ? f.second - b : 8)) - 1u});
    if (passes.empty()) return SPRS_HIP_OK;
    const uint64_t nchunks = (n + RS_CHUNK - 1) / RS_CHUNK;
    // temporaries: on the null stream from the library's pool, handed back in null-stream order (the pool's blocks may still be in
    // use by earlier null-stream work); on any other stream blocks of their own, freed behind that stream
    uint64_t *tk = nullptr, *tv = nullptr, *hist = nullptr, *offs = nullptr;
    uint64_t cap_tk = 0, cap_tv = 0, cap_hist = 0, cap_offs = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    const bool pooled = stream == nullptr;
    auto take = [&](uint64_t **p, uint64_t bytes, uint64_t *cap) {
        return pooled ? pool_alloc((void **)p, bytes, cap, dev) : hipMalloc((void **)p, bytes);
    };
    auto drop = [&](uint64_t *p, uint64_t cap) {
        if (!p) return;
        if (pooled) pool_free(p, cap, dev, true);
        else (void)hipFree(p);
    };
    auto cleanup = [&]() {
        if (!pooled) (void)hipStreamSynchronize(stream);
        drop(tk, cap_tk);
        drop(tv, cap_tv);
        drop(hist, cap_hist);
        drop(offs, cap_offs);
    };
    if (e == hipSuccess) e = take(&tk, n * 8, &cap_tk);
    if (e == hipSuccess) e = take(&tv, n * 8, &cap_tv);
    if (e == hipSuccess) e = take(&hist, (RS_BINS * nchunks + 1) * 8, &cap_hist);
    if (e == hipSuccess) e = take(&offs, (RS_BINS * nchunks + 1) * 8, &cap_offs);
    if (e != hipSuccess) {
        cleanup();
        return fail_hip(e, "radix_sort_pairs");
    }
    uint64_t *ik = keys, *iv = vals, *ok = tk, *ov = tv;
    const dim3 grid((unsigned)((nchunks + RS_WAVES - 1) / RS_WAVES)), block(RS_BLOCK);
    int32_t st = SPRS_HIP_OK;
    for (const auto &p : passes) {
        hipLaunchKernelGGL(rs_hist_kernel, grid, block, 0, stream, (const uint64_t *)ik, n, p.first, p.second, nchunks, hist);
        st = exclusive_scan_u64(hist, offs, RS_BINS * nchunks, stream);
        if (st != SPRS_HIP_OK) break;
        hipLaunchKernelGGL(rs_scatter_kernel, grid, block, 0, stream, (const uint64_t *)ik, (const uint64_t *)iv, n, p.first, p.second,
                           nchunks, (const uint64_t *)offs, ok, ov);
        std::swap(ik, ok);
        std::swap(iv, ov);
    }
    if (st == SPRS_HIP_OK && ik != keys) {
        e = hipMemcpyAsync(keys, ik, n * 8, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(vals, iv, n * 8, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) st = fail_hip(e, "radix_sort_pairs copy back");
    }
    if (st == SPRS_HIP_OK) {
        e = hipGetLastError();
        if (e == hipSuccess && stream != nullptr) e = hipStreamSynchronize(stream);   // (the temporaries go back to the pool below, in null-stream order)
        if (e != hipSuccess) st = fail_hip(e, "radix_sort_pairs");
    }
    cleanup();
    return st;
}

}  // namespace sprs_hip
