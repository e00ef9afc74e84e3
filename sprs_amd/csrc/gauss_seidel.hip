// Device Gauss-Seidel: twin of gauss_seidel() in the reference's heat example (sprs/examples/heat.rs:103-139) —
// SURVEY §8 (f3), the second "caller that loops on the SpMV": every sweep is followed by the residual `&mat * &x - rhs`,
// which is the SpMV hot path (csmat.rs:2119-2158 -> prod::csr_mulacc_dense_colmaj).
//
// The sweep itself is a recurrence: row i reads, for every stored column c < i, the value row c has just been given IN THIS
// sweep, and for c > i the value of the previous sweep (heat.rs:113-131 updates x in place).  What can run side by side is
// fixed by the structure alone: level(i) = 1 + max level(c) over the stored c < i (0 without any).  Rows of one level do
// not read each other.  The plan (built once per handle) sorts the rows by level; the sweep kernel then is ONE launch:
//   * a wave draws the next 64 positions of that order from a counter (in order, so every row a wave can ever wait for has
//     been drawn by a wave that is already running: no wave waits for one that has not started — no co-residency needed);
//   * a lane owns one row and walks its entries in the reference's order, eight at a time: the eight x values are requested
//     together (previous iterate: plain loads; this sweep's values: 8-byte agent-scope loads of the NEXT iterate, which the
//     host filled with a bit pattern no arithmetic produces), the products are added in entry order as far as the values
//     have arrived, the rest is polled again;
//   * x[row] = (rhs[row] - sigma) / diag is published with one 8-byte agent-scope store: value and "ready" are the same
//     granule, so there is no flag to order against the data (MI355X_MICROARCH.md, hand-off by data-tagged granules).
// Same operands, same order, unfused multiply-add (-ffp-contract=off), IEEE division: x is bit-identical to the CPU sweep.
// Double buffered (the previous iterate must stay readable for the columns after the row), so a sweep also costs one
// memset of n doubles; the user's x_dev receives the last iterate.
//
// Bound: the chain of levels, one publish -> poll hop (~1 us) each — 8 191 levels for the 5-point Laplacian of a 4096 x 4096
// grid — not bandwidth (the matrix streams once per sweep: 1.6 GB = 0.2 ms).  DESIGN.md §4.5.
//
// Safety net: a lane that polls longer than GS_SPIN_LIMIT rounds raises a status word that every wave looks at, and the
// kernel ends with an error instead of hanging (cannot happen with a correct level order; it guards the order, not the data).
#include "common.hpp"

#include <cstring>
#include <vector>
#include "scan.hpp"

#include <cmath>
#include <vector>

namespace sprs_hip {

int32_t radix_sort_pairs(uint64_t *keys, uint64_t *vals, uint64_t n, const std::vector<std::pair<int, int>> &fields, hipStream_t stream);   // sort.hip

void GsPlan::release() {
    if (order) (void)hipFree(order);
    order = nullptr;
    built = false;
    nlevels = 0;
    no_diag_row = UINT64_MAX;
    chain_tried = 0;
    chain_ok = false;
}

namespace {

constexpr int GS_BLOCK = 256;
constexpr int GS_B = 8;                                       // entries of a row requested together
constexpr unsigned long long GS_PENDING = ~0ull;              // "row not swept yet" in the next iterate: memset 0xFF, a NaN no arithmetic yields
constexpr unsigned long long GS_QNAN = 0x7FF8000000000000ull;
constexpr uint32_t GS_SPIN_LIMIT = 1u << 22;
constexpr unsigned int GS_NO_DIAG = 1u, GS_TIMEOUT = 2u;
constexpr uint64_t SUM_CHUNK = 8192;

#ifdef SPRS_HIP_EMU
#define GS_TOUCH(v) ((void)(v))
#else
#define GS_TOUCH(v) asm volatile("" ::"v"(v))
#endif

__device__ __forceinline__ unsigned long long gs_peek(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the XCD (accelerator die: its own L2) this wave runs on, from the hardware register
__device__ __forceinline__ uint32_t gs_xcc_id() {
#ifdef SPRS_HIP_EMU
    return 0u;
#else
    return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // hwreg(HW_REG_XCC_ID, 0, 4)
#endif
}

// ONE_XCD: only workgroups that find themselves on XCD 0 take part.  All of them then share one L2, so a result can be
// published with a plain store (it stays in that L2, dirty) and the 8-byte L1-bypassing loads of the pollers are served
// from there instead of from the fabric: a shorter hop for matrices whose sweep is a long chain of narrow levels.  Which
// XCD a wave is on is read from the hardware (never assumed from blockIdx), so only speed depends on the dispatcher.
template <typename IDX, typename PTR, bool ONE_XCD>
__global__ __launch_bounds__(GS_BLOCK) void gs_sweep_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                            const double *__restrict__ data,
                                                            const uint32_t *__restrict__ order,
                                                            const double *__restrict__ x_old, unsigned long long *x_new,
                                                            const double *__restrict__ rhs, uint64_t n,
                                                            unsigned int *next_chunk, unsigned int *status, uint32_t max_naps) {
    const uint32_t lane = threadIdx.x & 63u;
    if (ONE_XCD && gs_xcc_id() != 0u) return;
    auto draw = [&]() -> uint64_t {
        unsigned int q = 0;
        if (lane == 0) q = atomicAdd(next_chunk, 1u);
        return (uint64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)q) * 64u;
    };
    for (uint64_t base = draw(); base < n; base = draw()) {
        const uint64_t pos = base + lane;
        bool done = pos >= n;
        uint32_t row = 0;
        uint64_t p = 0, end = 0;
        double b = 0.0;
        if (!done) {
            row = order[pos];
            p = (uint64_t)indptr[row];
            end = (uint64_t)indptr[row + 1];
            b = rhs[row];
        }
        double sigma = 0.0, diag = 0.0;
        bool has_diag = false;
        uint32_t col[GS_B];
        double val[GS_B], xv[GS_B];
        uint32_t nb = 0, pend = 0, spins = 0;
        bool loaded = false, more = true;
        while (more) {
            bool moved = false;
            if (!done) {
                if (!loaded) {                                 // the next (up to) eight entries of my row
                    nb = end - p < (uint64_t)GS_B ? (uint32_t)(end - p) : (uint32_t)GS_B;
#pragma unroll
                    for (int u = 0; u < GS_B; ++u)
                        if ((uint32_t)u < nb) {
                            col[u] = (uint32_t)indices[p + u];
                            val[u] = data[p + u];
                        }
                    p += nb;
                    // The column ids are needed NOW (they address the x loads below).  Without a use at this point the compiler
                    // waits for them in front of every single x load with a count that must hold on every path (vmcnt(1)): the
                    // x loads then go out one after the other instead of together.
                    uint32_t touch = 0;
#pragma unroll
                    for (int u = 0; u < GS_B; ++u)
                        if ((uint32_t)u < nb) touch |= col[u];
                    GS_TOUCH(touch);
                    // columns behind the row: the previous iterate, requested once; columns before it: to be polled
                    pend = 0;
#pragma unroll
                    for (int u = 0; u < GS_B; ++u)
                        if ((uint32_t)u < nb) {
                            if (col[u] > row) xv[u] = x_old[col[u]];
                            else if (col[u] < row) pend |= 1u << u;
                        }
                    loaded = true;
                    moved = true;
                }
                // the poll: this sweep's values that have not arrived yet, all requested together — kept short, it is what a
                // waiting wave executes over and over and what stands between a published value and its use
                unsigned long long bits[GS_B];
#pragma unroll
                for (int u = 0; u < GS_B; ++u)
                    if ((pend >> u) & 1u) bits[u] = gs_peek(x_new + col[u]);
#pragma unroll
                for (int u = 0; u < GS_B; ++u)
                    if (((pend >> u) & 1u) && bits[u] != GS_PENDING) {
                        xv[u] = __longlong_as_double((long long)bits[u]);
                        pend &= ~(1u << u);
                        moved = true;
                    }
                if (pend == 0u) {                              // every operand of the batch is here: add in entry order (heat.rs:117-123)
#pragma unroll
                    for (int u = 0; u < GS_B; ++u)
                        if ((uint32_t)u < nb) {
                            if (col[u] == row) {
                                diag = val[u];
                                has_diag = true;
                            } else {
                                const double prod = val[u] * xv[u];
                                sigma = sigma + prod;
                            }
                        }
                    loaded = false;
                    moved = true;
                    if (p == end) {
                        double xr = (b - sigma) / diag;         // heat.rs:128-130
                        unsigned long long out = (unsigned long long)__double_as_longlong(xr);
                        if (!has_diag) {                        // diag.unwrap() of None: the host turns this into an error
                            atomicOr(status, GS_NO_DIAG);
                            out = GS_QNAN;
                        }
                        if (out == GS_PENDING) out = GS_QNAN;   // (a NaN payload handed through from rhs / x: still a NaN, but not "pending")
                        __hip_atomic_store(x_new + row, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // agent scope in both modes: the pollers sit on other CUs
                        done = true;
                    }
                }
            }
            more = __ballot(!done) != 0ull;
            if (more) {
                // rounds in which NO lane of the wave got anywhere (a lane that has finished its row would otherwise count every
                // round its neighbours are still busy with theirs and raise the timeout on a healthy sweep)
                spins = __ballot(moved) != 0ull ? 0u : spins + 1u;
                unsigned int st = 0;
                if ((spins & 63u) == 63u) st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins > GS_SPIN_LIMIT) {
                    atomicOr(status, GS_TIMEOUT);
                    st = GS_TIMEOUT;
                }
                if (__ballot((st & GS_TIMEOUT) != 0u) != 0ull) return;
                if (__ballot(moved) == 0ull) {                         // nobody got anywhere: let the publishers run, and back off
                    uint32_t naps = spins < 8u ? 1u : spins < 32u ? 2u : spins < 128u ? 4u : 8u;
                    if (naps > max_naps) naps = max_naps;
                    for (uint32_t q = 0; q < naps; ++q) SPRS_POLL_PAUSE();
                }
            }
        }
    }
}

// residual sum of the reference's convergence test: sum_i (v_i - rhs_i), v = A x.  Fixed two-level tree (chunks of 8192:
// thread t adds elements t, t + 256, ... in order, then the lanes and the waves in a fixed shape): deterministic, rounded
// differently from ndarray's eight running sums.
__global__ __launch_bounds__(GS_BLOCK) void gs_resid_partial_kernel(const double *__restrict__ v, const double *__restrict__ rhs,
                                                                    uint64_t n, double *__restrict__ partial) {
    __shared__ double lds[GS_BLOCK / 64];
    const uint64_t lo = (uint64_t)blockIdx.x * SUM_CHUNK;
    const uint64_t hi = lo + SUM_CHUNK < n ? lo + SUM_CHUNK : n;
    double s = 0.0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += GS_BLOCK) {
        const double r = v[i] - rhs[i];
        s = s + r;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
        for (int w = 1; w < GS_BLOCK / 64; ++w) t = t + lds[w];
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(GS_BLOCK) void gs_resid_final_kernel(const double *__restrict__ partial, uint64_t nchunks,
                                                                  double *__restrict__ out) {
    __shared__ double lds[GS_BLOCK / 64];
    double s = 0.0;
    for (uint64_t i = threadIdx.x; i < nchunks; i += GS_BLOCK) s = s + partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
        for (int w = 1; w < GS_BLOCK / 64; ++w) t = t + lds[w];
        out[0] = t;
    }
}

// The level order ON THE DEVICE, once per handle: level(i) = 1 + max level(c) over the stored c < i, rows sorted by (level, row).
// The levels are themselves a recurrence over the rows, and the sweep's own trick computes them: the waves draw rows in
// natural order, a lane owns a row and polls the level word of every column c < row (0xFFFFFFFF = not there yet) with
// L1-bypassing loads; every row a wave can wait for has a smaller number, i.e. was drawn by a wave that already runs (or is a
// lower lane of the same wave, which the polling loop serves first) — no grid barrier, no deadlock.  Then one stable radix sort
// of (level, row).  Rounds 1 to 3 downloaded the structure and ran this recurrence serially on the host: 0.41 s for the 1.7e7
// rows of the 4096 x 4096 heat system.
constexpr unsigned int LV_PENDING = 0xFFFFFFFFu;

template <typename IDX, typename PTR>
__global__ __launch_bounds__(GS_BLOCK) void gs_level_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices, uint64_t n,
                                                            unsigned int *level, unsigned int *next_chunk, unsigned int *status,
                                                            unsigned int *top, unsigned long long *no_diag) {
    const uint32_t lane = threadIdx.x & 63u;
    auto draw = [&]() -> uint64_t {
        unsigned int q = 0;
        if (lane == 0) q = atomicAdd(next_chunk, 1u);
        return (uint64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)q) * 64u;
    };
    for (uint64_t base = draw(); base < n; base = draw()) {
        const uint64_t row = base + lane;
        bool done = row >= n, diag = false;
        uint64_t p = 0, end = 0, c = row;
        uint32_t lv = 0, spins = 0;
        if (!done) {
            p = (uint64_t)indptr[row];
            end = (uint64_t)indptr[row + 1];
            if (p < end) c = (uint64_t)indices[p];      // the column the lane waits for stays in a register between the rounds
        }
        // One dependency per lane and round.  A dependency on a row of this very wave (the left neighbour of a grid row, say) is
        // answered by a shuffle — through memory the 64 lanes of a chunk of a chain would be 64 round trips one after the other
        // (first version: 1.6 s for the 4096 x 4096 heat system, slower than the host pass it replaces) — everything else by an
        // L1-bypassing load of the level word.  Columns ascend inside a row: the first c >= row ends the row's dependencies.
        while (__ballot(!done) != 0ull) {
            bool moved = false;
            const bool want = !done && p < end;
            const bool in_wave = want && c < row && c >= base;
            const unsigned int mine = done ? lv : LV_PENDING;
            const unsigned int from_wave = (unsigned int)__shfl((int)mine, in_wave ? (int)(c - base) : (int)lane, 64);
            if (want) {
                if (c < row) {
                    const unsigned int l = in_wave ? from_wave : __hip_atomic_load(level + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (l != LV_PENDING) {
                        if (l + 1u > lv) lv = l + 1u;
                        ++p;
                        if (p < end) c = (uint64_t)indices[p];
                        moved = true;
                    }
                } else {
                    diag = c == row;
                    p = end;                      // the rest of the row lies above the diagonal
                    moved = true;
                }
            }
            if (!done && p == end) {
                __hip_atomic_store(level + row, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!diag) atomicMin(no_diag, (unsigned long long)row);
                done = true;
                moved = true;
            }
            if (__ballot(!done) == 0ull) break;
            spins = __ballot(moved) != 0ull ? 0u : spins + 1u;          // rounds in which no lane of the wave got anywhere
            unsigned int st = 0;
            if ((spins & 63u) == 63u) st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spins > GS_SPIN_LIMIT) {
                atomicOr(status, GS_TIMEOUT);
                st = GS_TIMEOUT;
            }
            if (__ballot((st & GS_TIMEOUT) != 0u) != 0ull) return;
            if (__ballot(moved) == 0ull) SPRS_POLL_PAUSE();
        }
        // the highest level of the chunk: ONE atomic per wave (one per row — 1.7e7 atomics on a single word — was the whole
        // second of the first version)
        uint32_t wmax = row < n ? lv : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_down((int)wmax, off, 64);
            wmax = o > wmax ? o : wmax;
        }
        if (lane == 0) atomicMax(top, wmax);
    }
}

__global__ void gs_level_keys_kernel(const unsigned int *__restrict__ level, uint64_t n, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = level[i];
    vals[i] = i;
}

__global__ void gs_order_kernel(const uint64_t *__restrict__ vals, uint64_t n, uint32_t *__restrict__ order) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (uint32_t)vals[i];
}

struct DevTmp {
    void *p = nullptr;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T>
    T *as() { return (T *)p; }
};

template <typename IDX, typename PTR>
int32_t gs_plan_build(sprs_hip_csmat *a) {
    GsPlan &pl = a->gs;
    const uint64_t n = a->rows;
    hipStream_t stream = nullptr;
    DevTmp level, words, keys, vals;
    SPRS_TRY_HIP(level.alloc(n * sizeof(unsigned int)));
    SPRS_TRY_HIP(words.alloc(64));
    SPRS_TRY_HIP(keys.alloc(n * 8));
    SPRS_TRY_HIP(vals.alloc(n * 8));
    SPRS_TRY_HIP(hipMemsetAsync(level.p, 0xFF, n * sizeof(unsigned int), stream));
    SPRS_TRY_HIP(hipMemsetAsync(words.p, 0, 64, stream));
    SPRS_TRY_HIP(hipMemsetAsync(words.as<unsigned int>() + 4, 0xFF, 8, stream));          // no_diag = UINT64_MAX
    unsigned int *w = words.as<unsigned int>();         // [0] chunks drawn, [1] status, [2] top level, [4..5] first row without a diagonal
    int ncu = 0, dev = 0;
    SPRS_TRY_HIP(hipGetDevice(&dev));
    SPRS_TRY_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (n) {
        uint64_t grid = (uint64_t)(ncu > 0 ? ncu : 1) * 4;
        const uint64_t need = (n + GS_BLOCK - 1) / GS_BLOCK;
        if (grid > need) grid = need;
        hipLaunchKernelGGL((gs_level_kernel<IDX, PTR>), dim3((unsigned)grid), dim3(GS_BLOCK), 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, n, level.as<unsigned int>(), w, w + 1, w + 2, (unsigned long long *)(w + 4));
        SPRS_TRY_HIP(hipGetLastError());
        hipLaunchKernelGGL(gs_level_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const unsigned int *)level.as<unsigned int>(), n,
                           keys.as<uint64_t>(), vals.as<uint64_t>());
        SPRS_TRY_HIP(hipGetLastError());
    }
    unsigned int hw[6] = {0, 0, 0, 0, 0, 0};
    SPRS_TRY_HIP(hipMemcpy(hw, words.p, sizeof(hw), hipMemcpyDeviceToHost));
    if (hw[1] & GS_TIMEOUT) SPRS_FAIL(SPRS_HIP_HIP_ERROR, "Gauss-Seidel plan: the level recurrence did not finish");
    const uint32_t top = hw[2];
    int bits = 1;
    while (bits < 32 && (top >> bits) != 0u) ++bits;
    SPRS_TRY(radix_sort_pairs(keys.as<uint64_t>(), vals.as<uint64_t>(), n, {{0, bits}}, stream));       // stable: rows ascending inside a level
    SPRS_TRY_HIP(hipMalloc((void **)&pl.order, (n ? n : 1) * sizeof(uint32_t)));
    if (n) {
        hipLaunchKernelGGL(gs_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint64_t *)vals.as<uint64_t>(), n, pl.order);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            pl.release();
            return fail_hip(e, "Gauss-Seidel level order");
        }
    }
    SPRS_TRY_HIP(hipStreamSynchronize(stream));
    unsigned long long nd = 0;
    memcpy(&nd, hw + 4, 8);
    pl.nlevels = n ? (uint64_t)top + 1 : 0;
    pl.no_diag_row = nd;
    pl.built = true;
    return SPRS_HIP_OK;
}

// ---- chains of consecutive rows: a systolic sweep for banded / grid systems (option gauss_seidel_chain) -------------------------------
// In level order a sweep costs one publish -> poll hop through the memory system per LEVEL (2.3 us; 8 188 levels on the 4096 x 4096 heat
// system).  When the matrix is a band — row i reads row i - 1 and row i - S, the 5-point stencil with S = the grid width — the hops
// can stay inside a workgroup.  A BAND is 64 S consecutive rows; lane l of the workgroup owns the chain of rows B + l S ... B + l S +
// S - 1 and walks it in order, one row per STEP, skewed: at global step T lane l sweeps its row t = T - l.  Its two band
// neighbours are then exactly one step old — row i - 1 is the lane's own previous row, row i - S is what lane l - 1 swept at step
// T - 1 — and come out of a two-row ring in LDS.  The GB_WAVES waves of the workgroup take the steps in turn (wave w: steps w, w + 8, ...):
// while seven waves sweep their steps, the eighth's loads for its next steps are in flight (index pair three turns ahead, entries
// two, operands one), and an LDS word hands the turn on.  Every other stored column c < i is polled in the next iterate like in the
// level-order kernel (lane 0's row above lives in the previous band, 64 steps ahead of it).  Same operands, same order, same
// arithmetic per row: the iterates stay bit-identical to the CPU sweep.
// The schedule must respect every dependency, so the plan CHECKS it (gs_band_check_kernel): rows of at most GB_E entries, and every
// stored c < i inside i's band must be swept at an earlier step than i; a matrix that fails keeps the level-order kernel.
constexpr int GB_WAVES = 8, GB_BLOCK = GB_WAVES * 64, GB_E = 8;

#ifdef SPRS_HIP_EMU
__device__ __forceinline__ uint32_t gb_lds_load(const uint32_t *p) { return *(const volatile uint32_t *)p; }
__device__ __forceinline__ void gb_lds_store(uint32_t *p, uint32_t v) { *(volatile uint32_t *)p = v; }
#else
typedef __attribute__((address_space(3))) volatile uint32_t gb_lds_vu32;
__device__ __forceinline__ uint32_t gb_lds_load(const uint32_t *p) { return *(const gb_lds_vu32 *)p; }
__device__ __forceinline__ void gb_lds_store(uint32_t *p, uint32_t v) { *(gb_lds_vu32 *)p = v; }
#endif

template <typename IDX, typename PTR>
__global__ void gs_band_check_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices, uint64_t n, uint64_t S,
                                     unsigned int *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p0 = (uint64_t)indptr[i], p1 = (uint64_t)indptr[i + 1];
    if (p1 - p0 > (uint64_t)GB_E) {
        atomicOr(bad, 1u);
        return;
    }
    const uint64_t band = 64 * S, B = i / band * band;
    const uint64_t Ti = (i - B) / S + (i - B) % S;                       // lane + step
    for (uint64_t p = p0; p < p1; ++p) {
        const uint64_t c = (uint64_t)indices[p];
        if (c >= i) break;
        if (c >= B && (c - B) / S + (c - B) % S >= Ti) atomicOr(bad, 2u);
    }
}

struct GbRow {                       // one row on its way to the sweep
    uint32_t col[GB_E];
    double val[GB_E];
    double b;
};

template <typename IDX, typename PTR>
__global__ __launch_bounds__(GB_BLOCK) void gs_band_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                           const double *__restrict__ data, const double *__restrict__ x_old,
                                                           unsigned long long *x_new, const double *__restrict__ rhs, uint64_t n,
                                                           uint64_t S, unsigned int *next_band, unsigned int *status, uint32_t dbg) {
    __shared__ double xs[2][64];
    __shared__ uint32_t step_done, band_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t nsteps = S + 63;
    for (;;) {
        if (tid == 0) {
            band_s = atomicAdd(next_band, 1u);
            gb_lds_store(&step_done, 0u);
        }
        __syncthreads();
        const uint64_t B = (uint64_t)band_s * 64u * S;
        if (B >= n) return;
        const uint64_t chain0 = B + (uint64_t)lane * S;
        auto row_of = [&](uint64_t T, bool &act) -> uint64_t {           // my row at global step T
            const uint64_t t = T - lane;
            act = T >= lane && t < S && chain0 + t < n;
            return act ? chain0 + t : 0ull;
        };
        // the pipeline of one wave: q3 = index pair of step T + 3 W, r2 = entries of step T + 2 W, r1 = entries and operands of T + W.
        // Everything that can be decided ahead of a row's step is decided in stage 3, two turns ahead, and travels as bit masks: the
        // step itself — the only thing on the band's critical path — is two LDS reads, eight select / multiply / add triples in entry
        // order and the division.  (First version: the step classified its entries itself, ~1 200 instructions of divergent code for a
        // wave that runs alone on its SIMD: 1.7 us per step, no faster than a level of the level-order kernel.)
        uint64_t q3s = 0, q3e = 0, q2s = 0, q2e = 0;
        GbRow r2, r1;
        double valc[GB_E], opc[GB_E], op1[GB_E], bc = 0.0, dg1 = 0.0, dgc = 0.0;
        uint32_t nb2 = 0, nb1 = 0, m1 = 0, mc = 0;     // masks, one bit per entry: 0-7 own previous row, 8-15 chain above, 16-23 not added (diagonal, past the row), 24-31 polled, 32: has a diagonal (bit 31 of the high half: kept in hd)
        bool hd1 = false, hdc = false;
        auto stage1 = [&](uint64_t T) {                                   // index pair
            bool act;
            const uint64_t r = row_of(T, act);
            q3s = (uint64_t)indptr[r];
            q3e = act ? (uint64_t)indptr[r + 1] : q3s;
        };
        auto stage2 = [&](uint64_t T) {                                   // entries and rhs of the row whose pair arrived
            bool act;
            const uint64_t r = row_of(T, act);
            nb2 = (uint32_t)(q2e - q2s);
#pragma unroll
            for (int u = 0; u < GB_E; ++u) {
                const uint64_t at = nb2 ? q2s + ((uint32_t)u < nb2 ? (uint64_t)u : 0ull) : 0ull;      // (unconditional loads)
                r2.col[u] = (uint32_t)indices[at];
                r2.val[u] = data[at];
            }
            r2.b = rhs[r];
        };
        auto stage3 = [&](uint64_t T) {                                   // operands and masks of the row whose entries arrived
            bool act;
            const uint64_t r = row_of(T, act);
            // eight UNCONDITIONAL loads through a selected pointer: the previous iterate for columns behind the row, this sweep's word
            // (value or GS_PENDING) for columns before it
            unsigned long long bits[GB_E];
#pragma unroll
            for (int u = 0; u < GB_E; ++u) {
                const uint32_t c = (uint32_t)u < nb1 ? r1.col[u] : (uint32_t)r;
                // (the two band neighbours come out of LDS at the step: their words in the next iterate are being written right now by
                // the neighbouring lanes — nothing to fetch there; the load goes to the previous iterate and is dropped)
                const bool nb_prev = (uint64_t)c + 1 == r && T > lane, nb_above = lane > 0 && (uint64_t)c + S == r;
                const unsigned long long *src = ((uint64_t)c > r || nb_prev || nb_above || (uint32_t)u >= nb1) ? (const unsigned long long *)(x_old + c)
                                                                                                              : (const unsigned long long *)(x_new + c);
                bits[u] = gs_peek(src);
            }
            uint32_t m = 0;
            dg1 = 0.0;
            hd1 = false;
#pragma unroll
            for (int u = 0; u < GB_E; ++u) {
                const uint64_t c = r1.col[u];
                const bool in = (uint32_t)u < nb1 && act;
                const bool is_dg = in && c == r;
                const bool prev = in && c + 1 == r && T > lane;
                const bool above = in && !prev && lane > 0 && c + S == r;
                if (is_dg) {
                    dg1 = r1.val[u];
                    hd1 = true;
                }
                m |= (prev ? 1u : 0u) << u;
                m |= (above ? 1u : 0u) << (8 + u);
                m |= ((!in || is_dg) ? 1u : 0u) << (16 + u);
                m |= ((in && c < r && !prev && !above) ? 1u : 0u) << (24 + u);
            }
            m1 = m;
            // A row of an EARLIER band (lane 0's row above) is waited for HERE, two turns before its use: the band then trails the band
            // above by the depth of this pipeline and finds the value in a register when its step comes (waiting at the use keeps the
            // band "just in time": every step then pays a poll's round trip).  Rows of this band are not waited for here — the wave
            // itself may be the one that sweeps them.
            if (__ballot((m >> 24) != 0u) != 0ull) {
#pragma unroll
                for (int u = 0; u < GB_E; ++u) {
                    if (((m >> (24 + u)) & 1u) && (uint64_t)r1.col[u] < B) {
                        uint32_t sp = 0;
                        while (bits[u] == GS_PENDING) {
                            SPRS_POLL_PAUSE();
                            bits[u] = gs_peek(x_new + r1.col[u]);
                            if (++sp > GS_SPIN_LIMIT) {
                                atomicOr(status, GS_TIMEOUT);
                                break;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GB_E; ++u) op1[u] = __longlong_as_double((long long)bits[u]);
        };
        uint32_t colc[GB_E];                                              // columns of the current row: only the rare poll at the use reads them
        auto rotate_to_current = [&]() {
#pragma unroll
            for (int u = 0; u < GB_E; ++u) {
                valc[u] = r1.val[u];
                colc[u] = r1.col[u];
                opc[u] = op1[u];
            }
            bc = r1.b;
            mc = m1;
            dgc = dg1;
            hdc = hd1;
        };
        // fill the pipeline for my first step T = wave
        stage1((uint64_t)wave);
        q2s = q3s; q2e = q3e;
        stage2((uint64_t)wave);
        r1 = r2; nb1 = nb2;
        stage3((uint64_t)wave);
        rotate_to_current();
        stage1((uint64_t)wave + GB_WAVES);
        q2s = q3s; q2e = q3e;
        stage2((uint64_t)wave + GB_WAVES);
        r1 = r2; nb1 = nb2;
        stage3((uint64_t)wave + GB_WAVES);
        stage1((uint64_t)wave + 2 * GB_WAVES);
        q2s = q3s; q2e = q3e;
        stage2((uint64_t)wave + 2 * GB_WAVES);
        stage1((uint64_t)wave + 3 * GB_WAVES);
        bool dead = false;
        for (uint64_t T = wave; T < nsteps; T += GB_WAVES) {
            // ---- my turn ----
            uint32_t spins = 0;
            while (gb_lds_load(&step_done) != (uint32_t)T) {
                SPRS_POLL_PAUSE();
                if (++spins > GS_SPIN_LIMIT) {
                    atomicOr(status, GS_TIMEOUT);
                    dead = true;
                    break;
                }
            }
            if (dead) break;
            wave_sync_lds();
            bool act;
            const uint64_t row = row_of(T, act);
            // rare: a polled operand (a row of this band outside the two band neighbours, or of an earlier band when the pipeline
            // was filled) that was not there two turns ago
            {
                uint32_t pend = 0;
#pragma unroll
                for (int u = 0; u < GB_E; ++u)
                    if (((mc >> (24 + u)) & 1u) && (unsigned long long)__double_as_longlong(opc[u]) == GS_PENDING) pend |= 1u << u;
                if (__ballot(pend != 0u) != 0ull) {
#pragma unroll
                    for (int u = 0; u < GB_E; ++u)
                        if ((pend >> u) & 1u) {
                            unsigned long long bits = GS_PENDING;
                            uint32_t sp = 0;
                            while (bits == GS_PENDING) {
                                bits = gs_peek(x_new + colc[u]);
                                if (++sp > GS_SPIN_LIMIT) {
                                    atomicOr(status, GS_TIMEOUT);
                                    break;
                                }
                            }
                            opc[u] = __longlong_as_double((long long)bits);
                        }
                }
            }
            const uint32_t prev = (uint32_t)((T + 1) & 1);                // ring row written at step T - 1
            const double own = xs[prev][lane], left = xs[prev][lane ? lane - 1 : 0];
            double sigma = 0.0;
#pragma unroll
            for (int u = 0; u < GB_E; ++u) {                              // entry order; an entry that is not added contributes +0.0 (a sum that starts at +0.0 never is -0.0)
                const double xv = ((mc >> u) & 1u) ? own : ((mc >> (8 + u)) & 1u) ? left : opc[u];
                const double prod = valc[u] * xv;
                const double term = ((mc >> (16 + u)) & 1u) ? 0.0 : prod;
                sigma = sigma + term;
            }
            if (act) {
                const double xr = (bc - sigma) / dgc;                     // heat.rs:128-130
                unsigned long long out = (unsigned long long)__double_as_longlong(xr);
                if (!hdc && !(DEVTOOLS && dbg)) {
                    atomicOr(status, GS_NO_DIAG);
                    out = GS_QNAN;
                }
                if (out == GS_PENDING) out = GS_QNAN;
                xs[T & 1][lane] = __longlong_as_double((long long)out);
                __hip_atomic_store(x_new + row, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            SPRS_LDS_FENCE();
            wave_sync_lds();
            if (lane == 0) gb_lds_store(&step_done, (uint32_t)T + 1u);
            // ---- rotate the pipeline: the loads issued a turn ago have had seven steps of the other waves to arrive ----
            if (DEVTOOLS && (dbg & 1u)) continue;                        // timing experiment (WRONG results): no pipeline work at all
            rotate_to_current();
            r1 = r2; nb1 = nb2;
            if (!(DEVTOOLS && (dbg & 2u))) stage3(T + 2 * GB_WAVES);     // timing experiment (WRONG results): no operand loads
            q2s = q3s; q2e = q3e;
            if (!(DEVTOOLS && (dbg & 4u))) stage2(T + 3 * GB_WAVES);
            if (!(DEVTOOLS && (dbg & 8u))) stage1(T + 4 * GB_WAVES);
        }
        __syncthreads();                                                 // the band is done (or the sweep is being abandoned)
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & GS_TIMEOUT) return;
    }
}

struct Work {
    double *buf = nullptr;
    unsigned int *words = nullptr;
    hipStream_t stream = nullptr;
    ~Work() {
        (void)hipStreamSynchronize(stream);     // an early return must not leave copies into this frame's variables in flight
        if (buf) (void)hipFree(buf);
        if (words) (void)hipFree(words);
    }
};

template <typename IDX, typename PTR>
int32_t gs_impl(sprs_hip_csmat *a, double *x, const double *rhs, uint64_t n, uint64_t max_iter, double eps,
                sprs_hip_gauss_seidel_info *info, hipStream_t stream) {
    // held for the whole solve: the level order (and the SpMV plan of the residual) must outlive every launch that reads them;
    // sprs_hip_csmat_refresh / _free on another thread wait (recursive: the SpMV of the residual locks again)
    std::lock_guard<std::recursive_mutex> lock(a->mu);
    if (!a->gs.built) SPRS_TRY((gs_plan_build<IDX, PTR>(a)));
    const uint32_t *order = a->gs.order;
    const uint64_t nlevels = a->gs.nlevels, no_diag_row = a->gs.no_diag_row;
    // ---- the band schedule, when the matrix fits one ----
    uint64_t chain_S = 0;
    {
        const int64_t opt = options().gauss_seidel_chain;
        uint64_t want = opt > 1 ? (uint64_t)opt : 0;
        // (auto = level order: the band schedule is correct and bit-identical but NOT faster as built — 1.6 us per step on the
        // 4096 x 4096 heat system against 2.3 us per level, and twice as many steps as levels once 64 bands trail each other:
        // 19.0 against 18.6 ms per sweep, profiles/r10z.  Its step alone takes 0.7 us; the rest is the workgroup's loads: 64 chains
        // x 27 loads per step, each a 128-byte line of a different region, i.e. ~220 KB per step through ONE CU's 64 B / clk path
        // from L2.  It wants its operands fetched a line at a time and kept across the steps that share the line; until then it is
        // an explicit choice, gauss_seidel_chain = S.)
        if (want > 1 && want < n) {
            if (a->gs.chain_tried != want) {
                DevTmp flag;
                SPRS_TRY_HIP(flag.alloc(4));
                SPRS_TRY_HIP(hipMemsetAsync(flag.p, 0, 4, stream));
                hipLaunchKernelGGL((gs_band_check_kernel<IDX, PTR>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const PTR *)a->indptr,
                                   (const IDX *)a->indices, n, want, flag.as<unsigned int>());
                SPRS_TRY_HIP(hipGetLastError());
                unsigned int bad = 0;
                SPRS_TRY_HIP(hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, stream));
                SPRS_TRY_HIP(hipStreamSynchronize(stream));
                a->gs.chain_tried = want;
                a->gs.chain_ok = bad == 0;
            }
            if (a->gs.chain_ok) chain_S = want;
            else if (opt > 1)
                SPRS_FAIL(SPRS_HIP_INVALID_ARG, "gauss_seidel_chain = %lld: the matrix does not fit a band schedule of this stride (a row of more than %d "
                          "entries, or a dependency inside a band that the skewed chains would sweep later)", (long long)opt, GB_E);
        } else if (opt > 1) {
            SPRS_FAIL(SPRS_HIP_INVALID_ARG, "gauss_seidel_chain = %lld: the stride must be smaller than the number of rows", (long long)opt);
        }
    }
    if (max_iter > 0 && no_diag_row != UINT64_MAX)
        SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Gauss-Seidel: row %llu has no stored diagonal entry (the reference's diag.unwrap() panics, heat.rs:127)",
                  (unsigned long long)no_diag_row);

    const uint64_t nchunks = (n + SUM_CHUNK - 1) / SUM_CHUNK;
    unsigned int sweep_words[2] = {0, 0};      // (declared before `w`: its destructor drains the stream that may still write them)
    double h_sum = 0.0;
    Work w;
    w.stream = stream;
    SPRS_TRY_HIP(hipMalloc((void **)&w.buf, (2 * n + nchunks + 8) * sizeof(double)));
    SPRS_TRY_HIP(hipMalloc((void **)&w.words, 64));
    double *other = w.buf, *v = other + n, *partial = v + n, *scal = partial + nchunks;
    unsigned int *next_chunk = w.words, *status = w.words + 1;

    auto residual_error = [&](const double *xc, double &err) -> int32_t {      // (&mat * &x - rhs).sum().sqrt()
        SPRS_TRY(spmv_f64(a, xc, v, false, stream));
        hipLaunchKernelGGL(gs_resid_partial_kernel, dim3((unsigned)nchunks), dim3(GS_BLOCK), 0, stream, v, rhs, n, partial);
        hipLaunchKernelGGL(gs_resid_final_kernel, dim3(1), dim3(GS_BLOCK), 0, stream, partial, nchunks, scal);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY_HIP(hipMemcpyAsync(&h_sum, scal, sizeof(double), hipMemcpyDeviceToHost, stream));
        SPRS_TRY_HIP(hipStreamSynchronize(stream));
        err = std::sqrt(h_sum);
        return SPRS_HIP_OK;
    };

    int ncu = 0, dev = 0;
    SPRS_TRY_HIP(hipGetDevice(&dev));
    SPRS_TRY_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    // few workgroups on purpose: a level of the heat system is 64 waves wide, and every wave beyond the ones at the front
    // only adds polls to the memory queues the hand-offs go through (profiles/r07a: 256 / 512 / 1024 / 2048 workgroups:
    // 20 / 27 / 40 / 58 ms per sweep)
    const uint64_t need = (n + GS_BLOCK - 1) / GS_BLOCK;
    uint64_t grid = options().gauss_seidel_blocks > 0 ? (uint64_t)options().gauss_seidel_blocks : (uint64_t)(ncu > 0 ? ncu : 1);
    if (grid > need) grid = need;
    // one XCD (option): hand-offs through one L2 for long chains of narrow levels
    // (measured: no gain — 20.2 against 19.3 ms on the 4096^2 heat system, profiles/r08d, r08e — so auto means every XCD)
    const bool one_xcd = options().gauss_seidel_xcd == 1;
    if (one_xcd && options().gauss_seidel_blocks == 0) grid = (uint64_t)(ncu > 8 ? ncu / 8 : 1) < need ? (uint64_t)(ncu > 8 ? ncu / 8 : 1) : need;
    const uint32_t max_naps = options().gauss_seidel_naps > 0 ? (uint32_t)options().gauss_seidel_naps : 1u;

    double error = 0.0;
    int32_t converged = 0;
    uint64_t iterations = max_iter;
    double *cur = x, *nxt = other;
    if (max_iter == 0) SPRS_TRY(residual_error(cur, error));                   // heat.rs:111: what Err(error) holds then
    for (uint64_t it = 0; it < max_iter; ++it) {
        SPRS_TRY_HIP(hipMemsetAsync(nxt, 0xFF, n * sizeof(double), stream));   // every row of the next iterate "pending"
        SPRS_TRY_HIP(hipMemsetAsync(w.words, 0, 64, stream));
        if (chain_S) {     // one workgroup per band, bands drawn in order
            uint64_t nbands = (n + 64 * chain_S - 1) / (64 * chain_S);
            if (nbands > (uint64_t)(ncu > 0 ? ncu : 1)) nbands = (uint64_t)(ncu > 0 ? ncu : 1);
            hipLaunchKernelGGL((gs_band_kernel<IDX, PTR>), dim3((unsigned)nbands), dim3(GB_BLOCK), 0, stream, (const PTR *)a->indptr,
                               (const IDX *)a->indices, (const double *)a->data, (const double *)cur, (unsigned long long *)nxt, rhs, n, chain_S,
                               next_chunk, status, DEVTOOLS ? (uint32_t)options().gauss_seidel_debug : 0u);
        } else if (one_xcd)       // 8 x the workgroups: an eighth of them lands on XCD 0 and stays
            hipLaunchKernelGGL((gs_sweep_kernel<IDX, PTR, true>), dim3((unsigned)(grid * 8)), dim3(GS_BLOCK), 0, stream, (const PTR *)a->indptr,
                               (const IDX *)a->indices, (const double *)a->data, order, (const double *)cur,
                               (unsigned long long *)nxt, rhs, n, next_chunk, status, max_naps);
        else
            hipLaunchKernelGGL((gs_sweep_kernel<IDX, PTR, false>), dim3((unsigned)grid), dim3(GS_BLOCK), 0, stream, (const PTR *)a->indptr,
                               (const IDX *)a->indices, (const double *)a->data, order, (const double *)cur,
                               (unsigned long long *)nxt, rhs, n, next_chunk, status, max_naps);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY_HIP(hipMemcpyAsync(sweep_words, w.words, sizeof(sweep_words), hipMemcpyDeviceToHost, stream));   // [0] chunks drawn, [1] status
        double *t = cur;
        cur = nxt;
        nxt = t;
        SPRS_TRY(residual_error(cur, error));                                  // (drains the stream: the two words are valid after it)
        const unsigned int st = sweep_words[1];
        // every position of the level order must have been drawn by some wave: in the one-XCD mode the workgroups that do not
        // find themselves on XCD 0 leave at once, and a device that puts none there would return the 0xFF fill as the iterate
        if ((uint64_t)sweep_words[0] * 64u * (chain_S ? chain_S : 1u) < n)
            SPRS_FAIL(SPRS_HIP_HIP_ERROR, "Gauss-Seidel sweep: only %llu of %llu rows were swept (no workgroup took part: gauss_seidel_xcd on a device without workgroups on XCD 0?)",
                      (unsigned long long)sweep_words[0] * 64ull, (unsigned long long)n);
        if (st & GS_TIMEOUT)
            SPRS_FAIL(SPRS_HIP_HIP_ERROR, "Gauss-Seidel sweep: a row waited for a value that was never published (level order broken)");
        if (st & GS_NO_DIAG) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Gauss-Seidel: a row has no stored diagonal entry");
        if (error < eps) {                                                     // heat.rs:134-136
            converged = 1;
            iterations = it;
            break;
        }
    }
    if (cur != x) SPRS_TRY_HIP(hipMemcpyAsync(x, cur, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    SPRS_TRY_HIP(hipStreamSynchronize(stream));
    if (info) {
        info->iterations = iterations;
        info->error = error;
        info->converged = converged;
        info->levels = nlevels;
    }
    return SPRS_HIP_OK;
}

}  // namespace

int32_t gauss_seidel_f64(sprs_hip_csmat *a, double *x, const double *rhs, uint64_t n, uint64_t max_iter, double eps,
                         sprs_hip_gauss_seidel_info *info, hipStream_t stream) {
    if (n >= (1ull << 32)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "Gauss-Seidel: 2^32 rows or more are not supported");
    if (n == 0) {                      // an empty sum: error = sqrt(0) = 0
        if (info) *info = sprs_hip_gauss_seidel_info{max_iter && !(0.0 < eps) ? max_iter : 0, 0.0, (max_iter && 0.0 < eps) ? 1 : 0, 0};
        return SPRS_HIP_OK;
    }
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return gs_impl<uint64_t, uint64_t>(a, x, rhs, n, max_iter, eps, info, stream);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return gs_impl<uint32_t, uint64_t>(a, x, rhs, n, max_iter, eps, info, stream);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return gs_impl<uint64_t, uint32_t>(a, x, rhs, n, max_iter, eps, info, stream);
    return gs_impl<uint32_t, uint32_t>(a, x, rhs, n, max_iter, eps, info, stream);
}

}  // namespace sprs_hip
