// Device helpers shared by the SpMV translation units (spmv.hip: plain and XCD-sliced plans;
// spmv_band.hip: banded plan): wave reduction, column relabelling by popularity class, x-line hash.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace sprs_hip {

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int BLOCK = 256;       // 4 waves
constexpr int WAVE = 64;
constexpr int NWAVES = BLOCK / WAVE;
constexpr int SEG_CHUNK = 2048;  // row boundaries staged per pass
constexpr uint32_t LONG_SEG = 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// ---------------------------------------------------------------------------
// column relabelling of the sliced plan
//
// An L2 line holds 16 consecutive x entries.  In the natural labelling a hub column shares its
// line with 15 columns of arbitrary (on R-MAT: up to 100x lower) popularity, so most of what the
// 4 MiB of an XCD's L2 caches is cold.  The plan therefore renumbers the columns by POPULARITY
// CLASS — half octaves of the count, most popular first, natural order inside a class (one stable
// counting-sort pass) — so that lines are homogeneous, and gathers x into that order at the start
// of every SpMV (one pass over x: ~0.05 ms at 10 M columns).  Measured on the R-MAT 10 M matrix
// with the columns relabelled up front: 1.83 -> 1.68 ms; with the K hottest columns merely moved
// to the front: no gain; with a RANDOM relabelling: 2.38 ms (profiles/r01z_spmv_column_labelling.txt).
// The entries keep their order inside the rows; which slice an entry falls into follows its label, so
// the 8 partial sums of a row group the products differently than without relabelling (rounding-level
// differences, same run-to-run determinism).
// ---------------------------------------------------------------------------
constexpr int RL_CHUNK = 1024;            // columns per wave in the counting sort
constexpr int RL_DIGITS = 64;             // half-octave classes: 2 floor(log2 c) + (next bit of c) + 1; 0 = never referenced

__device__ __forceinline__ uint32_t rl_digit(uint32_t count) {
    uint32_t cls = 0;
    if (count) {
        uint32_t e = 31u - (uint32_t)__clz(count);                    // floor(log2(count))
        if (e > 30u) e = 30u;
        const uint32_t half = e ? (count >> (e - 1u)) & 1u : 0u;
        cls = 2u * e + half + 1u;                                      // 1 .. 62
    }
    return 63u - cls;                                                  // most popular first
}

// Exact counts (RL_SAMPLE = 1).  The atomics on the hub columns serialise in L2 — counting the 3.2e8 entries
// of the R-MAT 10M matrix takes 29 ms, half of the plan build — but estimating the popularity from every 4th
// entry (8 ms) mis-bins the rarely used columns and costs the SpMV 3 % (1.454 vs 1.415 ms for the sliced
// kernel, profiles/r01z_spmv_column_labelling.txt): the plan is built once, the SpMV runs many times.
constexpr uint64_t RL_SAMPLE = 1;

template <typename IDX>
__global__ __launch_bounds__(256) void rl_count_kernel(const IDX *__restrict__ indices, uint64_t nnz,
                                                       uint32_t *__restrict__ cnt) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n = (nnz + RL_SAMPLE - 1) / RL_SAMPLE;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride)
        atomicAdd(&cnt[indices[q * RL_SAMPLE]], 1u);
}

// one wave per chunk of RL_CHUNK columns; lane d keeps the number of columns of digit d
static __global__ __launch_bounds__(256) void rl_hist_kernel(const uint32_t *__restrict__ cnt, uint64_t cols, uint64_t nchunks,
                                                      uint64_t *__restrict__ hist) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    if (chunk >= nchunks) return;
    uint32_t mine = 0;
    for (int it = 0; it < RL_CHUNK / WAVE; ++it) {
        const uint64_t j = chunk * RL_CHUNK + (uint64_t)it * WAVE + lane;
        const uint32_t d = j < cols ? rl_digit(cnt[j]) : 0xFFu;
        for (uint32_t q = 0; q < (uint32_t)RL_DIGITS; ++q) {
            const uint32_t c = (uint32_t)__popcll(__ballot(d == q));
            if (lane == q) mine += c;
        }
    }
    if (lane < (uint32_t)RL_DIGITS) hist[(uint64_t)lane * nchunks + chunk] = mine;
}

// base = exclusive scan of hist (digit-major): label = base[digit][chunk] + rank inside the chunk
static __global__ __launch_bounds__(256) void rl_rank_kernel(const uint32_t *__restrict__ cnt, uint64_t cols, uint64_t nchunks,
                                                      const uint64_t *__restrict__ base, uint32_t *__restrict__ perm) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    if (chunk >= nchunks) return;
    uint32_t next = lane < (uint32_t)RL_DIGITS ? (uint32_t)base[(uint64_t)lane * nchunks + chunk] : 0u;   // lane d: next label of digit d
    for (int it = 0; it < RL_CHUNK / WAVE; ++it) {
        const uint64_t j = chunk * RL_CHUNK + (uint64_t)it * WAVE + lane;
        const uint32_t d = j < cols ? rl_digit(cnt[j]) : 0xFFu;
        for (uint32_t q = 0; q < (uint32_t)RL_DIGITS; ++q) {
            const unsigned long long m = __ballot(d == q);
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)next, (int)q);
            if (d == q) perm[j] = b + (uint32_t)__popcll(m & below);
            if (lane == q) next += (uint32_t)__popcll(m);
        }
    }
}

[[maybe_unused]] static __global__ __launch_bounds__(256) void rl_permute_x_kernel(const double *__restrict__ x, const uint32_t *__restrict__ perm,
                                                           uint64_t cols, double *__restrict__ xp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < cols) xp[perm[j]] = x[j];
}

// slice of an x entry = top 3 bits of a multiplicative (Fibonacci) hash of its 128-byte
// line number.  A plain bit field ((col >> 4) & 7) is badly unbalanced on R-MAT, whose
// column bits are each 0 with probability .76 (slice 0 would get 44 % of the entries: the
// first sliced plan ran 2x slower than no slicing for exactly that reason); the hash is
// balanced to ~1 % on R-MAT scale 24 and spreads consecutive lines of banded matrices.
__device__ __forceinline__ uint32_t x_slice(uint64_t col) {
    return (uint32_t)(((col >> 4) * 0x9E3779B97F4A7C15ull) >> 61);
}

struct TmpBuf {
    void *p = nullptr;
    ~TmpBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    uint64_t *u64() { return (uint64_t *)p; }
};


// perm[col] = label of the column: popularity classes, most referenced first (see the rl_* kernels).
// Allocates *perm (cols entries); the caller owns it.
template <typename IDX>
static int32_t build_column_labels(const IDX *indices, uint64_t nnz, uint64_t cols, hipStream_t stream, uint32_t **perm,
                                   uint64_t *nref = nullptr) {
    const uint64_t nchunks = (cols + RL_CHUNK - 1) / RL_CHUNK;
    TmpBuf ccount, hist, base;
    SPRS_TRY_HIP(ccount.alloc(cols * 4));
    SPRS_TRY_HIP(hist.alloc((RL_DIGITS * nchunks + 1) * 8));
    SPRS_TRY_HIP(base.alloc((RL_DIGITS * nchunks + 1) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)perm, cols * sizeof(uint32_t)));
    SPRS_TRY_HIP(hipMemsetAsync(ccount.p, 0, cols * 4, stream));
    hipLaunchKernelGGL(rl_count_kernel<IDX>, dim3(256 * 16), dim3(256), 0, stream, indices, nnz, (uint32_t *)ccount.p);
    SPRS_TRY_HIP(hipGetLastError());
    const dim3 wgrid((unsigned)((nchunks + 3) / 4));
    hipLaunchKernelGGL(rl_hist_kernel, wgrid, dim3(256), 0, stream, (const uint32_t *)ccount.p, cols, nchunks, hist.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(hist.u64(), base.u64(), RL_DIGITS * nchunks, stream));
    hipLaunchKernelGGL(rl_rank_kernel, wgrid, dim3(256), 0, stream, (const uint32_t *)ccount.p, cols, nchunks,
                       (const uint64_t *)base.u64(), *perm);
    SPRS_TRY_HIP(hipGetLastError());
    // the columns nobody references are the last class: its first label = the number of referenced columns
    if (nref) SPRS_TRY_HIP(hipMemcpy(nref, base.u64() + (uint64_t)(RL_DIGITS - 1) * nchunks, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipStreamSynchronize(stream));   // the temporaries go away here
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
