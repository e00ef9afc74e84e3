// Hand-written two-level exclusive prefix sum over uint64 — the device twin of
// the serial loop that turns per-row counts into the output indptr in
// smmp::mul_csr_csr_with_workspace (sprs/src/sparse/smmp.rs:320-331); also used
// to build the SpMV plans.
//   out[i] = sum_{j<i} in[j]  for i = 0..n   (out has n+1 entries, out[n] = total)
#include "common.hpp"
#include "scan.hpp"

namespace sprs_hip {

namespace {
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_BLOCK) void scan_partial_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                                                  uint64_t *__restrict__ sums) {
    __shared__ uint64_t wt[16];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += in[base + i];
    uint64_t tot;
    (void)block_excl_scan_u64(s, wt, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// exclusive scan of the block sums in place; sums[nblocks] receives the grand total
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint64_t *__restrict__ sums, uint64_t nblocks) {
    __shared__ uint64_t wt[16];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nblocks ? sums[i] : 0;
        uint64_t tot;
        const uint64_t ex = block_excl_scan_u64(v, wt, &tot);
        if (i < nblocks) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) sums[nblocks] = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_final_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                                                const uint64_t *__restrict__ sums, uint64_t nblocks,
                                                                uint64_t *__restrict__ out) {
    __shared__ uint64_t wt[16];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    uint64_t tot;
    uint64_t run = sums[blockIdx.x] + block_excl_scan_u64(s, wt, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nblocks];
}
}  // namespace

int32_t exclusive_scan_u64(const uint64_t *in, uint64_t *out, uint64_t n, hipStream_t stream) {
    if (n == 0) {
        SPRS_TRY_HIP(hipMemsetAsync(out, 0, sizeof(uint64_t), stream));
        return SPRS_HIP_OK;
    }
    const uint64_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    // block sums: on the null stream from the library's pool, handed back in null-stream order (pool_free, common.hpp: the pool's
    // blocks may still be in use by earlier null-stream work); on any other stream a block of its own, freed behind that stream
    uint64_t *sums = nullptr, cap = 0;
    int dev = 0;
    SPRS_TRY_HIP(hipGetDevice(&dev));
    if (stream == nullptr) SPRS_TRY_HIP(pool_alloc((void **)&sums, (nblocks + 1) * sizeof(uint64_t), &cap, dev));
    else SPRS_TRY_HIP(hipMalloc((void **)&sums, (nblocks + 1) * sizeof(uint64_t)));
    hipLaunchKernelGGL(scan_partial_kernel, dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, in, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nblocks);
    hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, in, n, sums, nblocks, out);
    hipError_t e = hipGetLastError();
    if (stream == nullptr) {
        pool_free(sums, cap, dev, true);
    } else {
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(sums);
    }
    if (e != hipSuccess) return fail_hip(e, "exclusive_scan_u64");
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
