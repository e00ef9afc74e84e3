// CSR x CSR SpGEMM for gfx950 — device twin of smmp::mul_csr_csr
// (sprs/src/sparse/smmp.rs:196-416): symbolic (smmp.rs:81-131) -> prefix sum of
// the per-row counts (smmp.rs:320-331) -> numeric (smmp.rs:151-189).
//
// Output contract (what makes indptr/indices bit-exact with the reference):
// every column reachable through A_i -> B_k is emitted, no numeric test
// (structural zeros kept, smmp.rs:109-119), and each row is strictly increasing
// (sort_unstable, smmp.rs:126).  Values: every C(i,j) is accumulated from +0.0
// over k in ascending order with a separately rounded multiply and add
// (smmp.rs:174-181, mul_acc.rs:28-30) — the same order as the reference, by a
// single owner, so results are deterministic (no float atomics anywhere).
//
// Work decomposition: a TASK is (row i, column window w).
//   * rows whose product count  ub_i = sum_{k in A_i} nnz(B_k)  is <= 512 are one
//     task handled by ONE WAVE with an LDS hash table (keys + f64 accumulators),
//     then a bitonic sort of the table in LDS;
//   * larger rows get one task per column window — 2^19 columns, narrowed down to 2^13 for
//     heavy rows so that a hub row becomes many tasks — handled by a 512-thread workgroup
//     with an LDS BITMAP of the window (64 KiB at most): setting bits is the
//     symbolic pass, a popcount prefix over the bitmap turns a column into its
//     rank inside the (sorted!) output row, so indices come out sorted for free.
//     Values are accumulated in LDS, in passes of <= 6144 outputs (superblock ranges of
//     the window); within a pass every wave owns a product-balanced range of 4096-column
//     superblocks and applies the k's in ascending order (8 k's prefetched at a time).
//   * a column-bucket table of B (entries of every row before each 4096-column boundary,
//     built per call, 4 B per row per 4096 columns) replaces the binary searches that
//     locate a row inside a window / superblock range, and yields the per-superblock
//     product counts used for the balancing.
// Per-task counts are scanned (hand-written two-level prefix sum, scan.hip) into output
// offsets; C.indptr falls out of the same scan.  Integer/LDS/HBM-bound: no MFMA.
// History of what was measured (profiles/): per-row windows + LDS accumulators 1.96 s ->
// 0.50 s on config 5; bucket table, product balancing, LDS staging of the k metadata,
// LDS-only passes and 8-deep prefetch together -> 0.465 s; a flattened (load-balanced)
// entry walk was slower and was dropped.  SPGEMM_PROF (option spgemm_prof) prints the
// phase profile of the large-row numeric kernel.
#include "common.hpp"
#include "scan.hpp"

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint64_t SMALL_MAX = 512;       // products per row handled by the wave/hash path
constexpr int SMALL_TAB = 1024;           // hash slots per wave (load factor <= 0.5)
constexpr int SM_BLOCK = 256;             // 4 waves
constexpr int SM_WAVES = SM_BLOCK / WAVE;
constexpr int WIN_LOG2 = 19;              // columns per window
constexpr uint64_t WIN = 1ull << WIN_LOG2;
constexpr int WORDS = (int)(WIN / 64);    // 8192 64-bit words = 64 KiB
constexpr int LG_BLOCK = 512;             // 8 waves
constexpr int LG_WAVES = LG_BLOCK / WAVE;
constexpr int WORDS_PER_THREAD = WORDS / LG_BLOCK;   // 16
constexpr int SUPER_WORDS = 64;           // words per superblock (4096 columns)
constexpr int NSUPER = WORDS / SUPER_WORDS;          // 128
constexpr int MIN_WIN_LOG2 = 13;          // heavy rows: windows down to 8192 columns
constexpr int ACC_CAP = 6144;             // tasks with at most this many outputs accumulate in LDS (48 KiB)
constexpr int K_CAP = 256;                // k's whose metadata is staged in LDS per pass

// first position in [lo, hi) whose column is >= v
template <typename IDX>
__device__ __forceinline__ uint64_t lower_bound_col(const IDX *__restrict__ idx, uint64_t lo, uint64_t hi, uint64_t v) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)idx[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <typename IDX, typename PTR>
struct CsrView {
    const PTR *indptr;
    const IDX *indices;
    const double *data;
    // optional column-bucket table of the right operand: bucket[k * nb + b] = number of entries of
    // row k with column < b * 4096.  Turns "where does row k enter column window [lo,hi)" — two
    // binary searches, ~20 dependent loads per (row, window) — into two independent loads.
    const uint32_t *bucket;
    uint64_t nb;
};

constexpr int BUCKET_LOG2 = 12;   // = one superblock of the LDS bitmap (64 words x 64 columns)

// sub-range [s,e) of row k (given its [s,e) = whole row) inside columns [lo, hi); lo is a multiple
// of 4096 and hi is either a multiple of 4096 or the number of columns
template <typename IDX, typename PTR>
__device__ __forceinline__ void row_window(const CsrView<IDX, PTR> &B, uint64_t k, uint64_t lo, uint64_t hi,
                                           uint64_t &s, uint64_t &e) {
    if (B.bucket) {
        const uint32_t *t = B.bucket + k * B.nb;
        const uint64_t row0 = s;
        s = row0 + t[lo >> BUCKET_LOG2];
        e = row0 + t[(hi + ((1ull << BUCKET_LOG2) - 1)) >> BUCKET_LOG2];
    } else {
        s = lower_bound_col(B.indices, s, e, lo);
        e = lower_bound_col(B.indices, s, e, hi);
    }
}

// bucket table build: one wave per row
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void build_bucket_kernel(const PTR *__restrict__ indptr,
                                                           const IDX *__restrict__ indices, uint64_t rows,
                                                           uint64_t nb, uint32_t *__restrict__ bucket) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x / 64);
    for (uint64_t r = w0; r < rows; r += nw) {
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        uint32_t *t = bucket + r * nb;
        // entry q (first of its bucket b, previous entry in bucket pb < b) defines t[pb+1 .. b] = q
        for (uint64_t p = s + lane; p < e; p += 64) {
            const uint64_t b = (uint64_t)indices[p] >> BUCKET_LOG2;
            const int64_t pb = p > s ? (int64_t)((uint64_t)indices[p - 1] >> BUCKET_LOG2) : -1;
            for (int64_t bb = pb + 1; bb <= (int64_t)b; ++bb) t[bb] = (uint32_t)(p - s);
        }
        const int64_t lastb = e > s ? (int64_t)((uint64_t)indices[e - 1] >> BUCKET_LOG2) : -1;
        for (uint64_t bb = (uint64_t)(lastb + 1) + lane; bb < nb; bb += 64) t[bb] = (uint32_t)(e - s);
    }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// value of `v` in lane `src`, `src` wave-uniform: v_readlane (a few cycles) instead of the
// ds_bpermute (an LDS round trip, ~100+ cycles) that __shfl with a variable index compiles to
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), src));
}

__device__ __forceinline__ uint32_t hash_slot(uint32_t c, int lg) { return (c * 0x9E3779B1u) >> (32 - lg); }

__device__ __forceinline__ int ceil_log2_u32(uint32_t v) { return v <= 1 ? 0 : 32 - __clz(v - 1); }

// ---------------------------------------------------------------------------
// pass 0: per-row product count and number of tasks
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void row_work_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B, uint64_t rows,
                                                       uint64_t b_cols, uint64_t heavy_products,
                                                       uint64_t *__restrict__ ub, uint64_t *__restrict__ ntasks,
                                                       uint8_t *__restrict__ wlog) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x / WAVE);
    for (uint64_t r = w0; r < rows; r += nw) {
        const uint64_t s = (uint64_t)A.indptr[r], e = (uint64_t)A.indptr[r + 1];
        uint64_t acc = 0;
        for (uint64_t p = s + lane; p < e; p += WAVE) {
            const uint64_t k = (uint64_t)A.indices[p];
            acc += (uint64_t)B.indptr[k + 1] - (uint64_t)B.indptr[k];
        }
        acc = wave_sum_u64(acc);
        if (lane == 0) {
            ub[r] = acc;
            // Window width of a large row: 2^19 columns, narrowed (down to 2^13) for heavy rows so that
            // a hub row becomes many tasks of ~HEAVY_PRODUCTS products instead of one serial chain.
            uint32_t wl = WIN_LOG2;
            if (acc > SMALL_MAX) {
                uint64_t want = acc / heavy_products;          // desired number of tasks
                if (want < 1) want = 1;
                uint64_t width = b_cols / want;                // columns per task
                wl = width <= 1 ? 0 : 63 - __clzll((long long)width);   // floor(log2)
                if (wl > (uint32_t)WIN_LOG2) wl = WIN_LOG2;
                if (wl < (uint32_t)MIN_WIN_LOG2) wl = MIN_WIN_LOG2;
            }
            wlog[r] = (uint8_t)wl;
            const uint64_t width = 1ull << wl;
            ntasks[r] = acc == 0 ? 0 : (acc <= SMALL_MAX ? 1 : (b_cols + width - 1) / width);
        }
    }
}

__global__ void make_tasks_kernel(const uint64_t *__restrict__ ub, const uint64_t *__restrict__ ntasks,
                                  const uint64_t *__restrict__ first_task, uint64_t rows,
                                  uint64_t *__restrict__ task_row, uint64_t *__restrict__ small_list,
                                  uint64_t *__restrict__ large_list, unsigned long long *__restrict__ counters) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t n = ntasks[r];
    if (!n) return;
    const uint64_t f = first_task[r];
    for (uint64_t j = 0; j < n; ++j) task_row[f + j] = r;
    if (ub[r] <= SMALL_MAX) {
        small_list[atomicAdd(&counters[0], 1ull)] = f;
    } else {
        const uint64_t pos = atomicAdd(&counters[1], (unsigned long long)n);
        for (uint64_t j = 0; j < n; ++j) large_list[pos + j] = f + j;
    }
}

// ---------------------------------------------------------------------------
// small rows: one wave per task, LDS hash table
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR, bool NUMERIC>
__global__ __launch_bounds__(SM_BLOCK) void small_rows_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B,
                                                              const uint64_t *__restrict__ small_list,
                                                              uint64_t n_small, const uint64_t *__restrict__ task_row,
                                                              const uint64_t *__restrict__ ub,
                                                              uint64_t *__restrict__ count,        // symbolic: out
                                                              const uint64_t *__restrict__ off,    // numeric: in
                                                              IDX *__restrict__ c_indices, double *__restrict__ c_data) {
    __shared__ uint32_t keys_s[SM_WAVES][SMALL_TAB];
    __shared__ double vals_s[NUMERIC ? SM_WAVES : 1][NUMERIC ? SMALL_TAB : 1];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    uint32_t *keys = keys_s[wave];
    double *vals = vals_s[NUMERIC ? wave : 0];
    const uint64_t w0 = (uint64_t)blockIdx.x * SM_WAVES + wave;
    const uint64_t nw = (uint64_t)gridDim.x * SM_WAVES;
    for (uint64_t q = w0; q < n_small; q += nw) {
        const uint64_t t = small_list[q];
        const uint64_t r = task_row[t];
        // table size: symbolic sizes it by the product count, numeric by the exact row count
        const uint32_t need = NUMERIC ? (uint32_t)count[t] : (uint32_t)ub[r];
        int lg = ceil_log2_u32(2 * need);
        if (lg < 6) lg = 6;
        const uint32_t tsize = 1u << lg, mask = tsize - 1;
        for (uint32_t i = lane; i < tsize; i += WAVE) keys[i] = EMPTY;
        __builtin_amdgcn_wave_barrier();
        const uint64_t as = (uint64_t)A.indptr[r], ae = (uint64_t)A.indptr[r + 1];
        uint32_t fresh = 0;
        for (uint64_t p0 = as; p0 < ae; p0 += WAVE) {
            const uint64_t p = p0 + lane;
            const bool valid = p < ae;
            const uint64_t k = valid ? (uint64_t)A.indices[p] : 0;
            const double av = valid ? A.data[p] : 0.0;
            const uint64_t bs = valid ? (uint64_t)B.indptr[k] : 0, be = valid ? (uint64_t)B.indptr[k + 1] : 0;
            const int nb = (ae - p0 < (uint64_t)WAVE) ? (int)(ae - p0) : WAVE;
            for (int j = 0; j < nb; ++j) {            // k ascending: the reference's order (smmp.rs:174-181)
                const uint64_t bsj = __shfl(bs, j, WAVE), bej = __shfl(be, j, WAVE);
                const double avj = __shfl(av, j, WAVE);
                for (uint64_t b = bsj + lane; b < bej; b += WAVE) {
                    const uint32_t c = (uint32_t)B.indices[b];
                    double pr = 0.0;
                    if constexpr (NUMERIC) pr = avj * B.data[b];
                    uint32_t h = hash_slot(c, lg);
                    for (;;) {
                        const uint32_t old = atomicCAS(&keys[h], EMPTY, c);
                        if (old == EMPTY) {
                            ++fresh;
                            if constexpr (NUMERIC) vals[h] = 0.0 + pr;     // tmp starts at N::zero()
                            break;
                        }
                        if (old == c) {
                            if constexpr (NUMERIC) vals[h] += pr;          // columns of one B row are distinct
                            break;
                        }
                        h = (h + 1) & mask;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (!NUMERIC) {
            const uint64_t tot = wave_sum_u64(fresh);
            if (lane == 0) count[t] = tot;
        } else {
            // bitonic sort of the table by key (EMPTY sorts last), values follow
            for (uint32_t k2 = 2; k2 <= tsize; k2 <<= 1) {
                for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = lane; i < tsize; i += WAVE) {
                        const uint32_t l = i ^ j;
                        if (l > i) {
                            const uint32_t ki = keys[i], kl = keys[l];
                            const bool asc = (i & k2) == 0;
                            if ((ki > kl) == asc) {
                                keys[i] = kl;
                                keys[l] = ki;
                                const double vi = vals[i], vl = vals[l];
                                vals[i] = vl;
                                vals[l] = vi;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            const uint64_t o = off[t];
            for (uint32_t i = lane; i < need; i += WAVE) {
                c_indices[o + i] = (IDX)keys[i];
                c_data[o + i] = vals[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------
// large rows: one workgroup per (row, column window) task, LDS bitmap
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR>
__device__ __forceinline__ void set_window_bits(const CsrView<IDX, PTR> &A, const CsrView<IDX, PTR> &B, uint64_t as,
                                                uint64_t ae, uint64_t wlo, uint64_t whi, bool whole_row,
                                                unsigned long long *bm, uint32_t &fresh) {
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    for (uint64_t p0 = as + (uint64_t)wave * WAVE; p0 < ae; p0 += (uint64_t)LG_WAVES * WAVE) {
        const uint64_t p = p0 + lane;
        const bool valid = p < ae;
        const uint64_t k = valid ? (uint64_t)A.indices[p] : 0;
        uint64_t s = valid ? (uint64_t)B.indptr[k] : 0, e = valid ? (uint64_t)B.indptr[k + 1] : 0;
        if (!whole_row && e > s) row_window(B, k, wlo, whi, s, e);
        unsigned long long live = __ballot(e > s);
        constexpr int PF = 8;                       // steps loaded together (see large_numeric_kernel)
        while (live) {
            int js[PF];
            int n = 0;
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                js[d] = 0;
                if (live) {
                    js[d] = __ffsll((long long)live) - 1;
                    live &= live - 1;
                    n = d + 1;
                }
            }
            uint64_t sj[PF], ej[PF], cc[PF];
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                sj[d] = readlane_u64(s, js[d]);
                ej[d] = readlane_u64(e, js[d]);
                cc[d] = 0;
                if (d < n && sj[d] + lane < ej[d]) cc[d] = (uint64_t)B.indices[sj[d] + lane];
            }
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                if (d < n) {
                    for (uint64_t b = sj[d] + lane; b < ej[d]; b += WAVE) {
                        const uint64_t c = (b == sj[d] + lane ? cc[d] : (uint64_t)B.indices[b]) - wlo;
                        const unsigned long long bit = 1ull << (c & 63);
                        const unsigned long long old = atomicOr(&bm[c >> 6], bit);
                        fresh += (old & bit) ? 0u : 1u;
                    }
                }
            }
        }
    }
}

template <typename IDX, typename PTR>
__global__ __launch_bounds__(LG_BLOCK) void large_symbolic_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B,
                                                                  uint64_t b_cols, const uint64_t *__restrict__ large_list,
                                                                  const uint64_t *__restrict__ task_row,
                                                                  const uint64_t *__restrict__ first_task,
                                                                  const uint64_t *__restrict__ ntasks,
                                                                  const uint8_t *__restrict__ wlog,
                                                                  uint64_t *__restrict__ count) {
    __shared__ unsigned long long bm[WORDS];
    __shared__ uint64_t red[LG_WAVES];
    const uint64_t t = large_list[blockIdx.x];
    const uint64_t r = task_row[t];
    const uint64_t w = t - first_task[r];
    const uint32_t wl = wlog[r];
    const int words = (int)((1ull << wl) / 64);
    const uint64_t wlo = w << wl, whi = (wlo + (1ull << wl) < b_cols) ? wlo + (1ull << wl) : b_cols;
    for (int i = threadIdx.x; i < words; i += LG_BLOCK) bm[i] = 0;
    __syncthreads();
    uint32_t fresh = 0;
    set_window_bits(A, B, (uint64_t)A.indptr[r], (uint64_t)A.indptr[r + 1], wlo, whi, ntasks[r] == 1, bm, fresh);
    const uint64_t ws = wave_sum_u64(fresh);
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tot = 0;
        for (int i = 0; i < LG_WAVES; ++i) tot += red[i];
        count[t] = tot;
    }
}

template <typename IDX, typename PTR>
__global__ __launch_bounds__(LG_BLOCK) void large_numeric_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B,
                                                                 uint64_t b_cols, const uint64_t *__restrict__ large_list,
                                                                 const uint64_t *__restrict__ task_row,
                                                                 const uint64_t *__restrict__ first_task,
                                                                 const uint64_t *__restrict__ ntasks,
                                                                 const uint8_t *__restrict__ wlog,
                                                                 const uint64_t *__restrict__ count,
                                                                 const uint64_t *__restrict__ off,
                                                                 IDX *__restrict__ c_indices, double *__restrict__ c_data,
                                                                 unsigned long long *__restrict__ prof) {
    __shared__ unsigned long long bm[WORDS];        // 64 KiB
    __shared__ uint16_t sub[WORDS];                 // 16 KiB: rank of a word inside its superblock
    __shared__ uint32_t super[NSUPER + 1];          // outputs before each 4096-column superblock
    __shared__ double acc[ACC_CAP];                 // 32 KiB: accumulators of tasks with few outputs
    __shared__ uint64_t wt[16];
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const uint64_t t = large_list[blockIdx.x];
    const uint64_t r = task_row[t];
    const uint64_t w = t - first_task[r];
    const uint32_t wl = wlog[r];
    const int words = (int)((1ull << wl) / 64);
    const uint64_t wlo = w << wl, whi = (wlo + (1ull << wl) < b_cols) ? wlo + (1ull << wl) : b_cols;
    const uint64_t as = (uint64_t)A.indptr[r], ae = (uint64_t)A.indptr[r + 1];
    const uint64_t out = off[t];
    const uint32_t cnt = (uint32_t)count[t];
    if (cnt == 0) return;                           // window without outputs (block-uniform)
    // accumulators: LDS when the task has few outputs, and ALWAYS with the bucket table (a wide
    // window is then processed in passes of <= ACC_CAP outputs); else the row's slots in L2
    const bool in_lds = cnt <= (uint32_t)ACC_CAP || B.bucket != nullptr;   // block-uniform
    uint32_t base_rank = 0;                          // first output of the current pass
    // optional phase profile (debug option spgemm_prof): cycles of thread 0 per phase, summed over tasks
    long long t_prev = prof ? (long long)clock64() : 0;
    auto mark = [&](int phase) {
        if (prof && tid == 0) {
            const long long now = (long long)clock64();
            atomicAdd(&prof[phase], (unsigned long long)(now - t_prev));
            t_prev = now;
        }
    };

    for (int i = tid; i < words; i += LG_BLOCK) bm[i] = 0;
    if (in_lds && !B.bucket)
        for (uint32_t i = tid; i < cnt; i += LG_BLOCK) acc[i] = 0.0;
    __syncthreads();
    uint32_t fresh = 0;
    set_window_bits(A, B, as, ae, wlo, whi, ntasks[r] == 1, bm, fresh);
    __syncthreads();
    mark(0);   // clear + bits

    // popcount prefix: thread tid owns words [16 tid, 16 tid + 16)
    uint32_t local[WORDS_PER_THREAD];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < WORDS_PER_THREAD; ++i) {
        const int word = tid * WORDS_PER_THREAD + i;
        local[i] = mine;
        mine += word < words ? (uint32_t)__popcll(bm[word]) : 0u;
    }
    uint64_t tot;
    const uint32_t tpre = (uint32_t)block_excl_scan_u64(mine, wt, &tot);
    constexpr int THREADS_PER_SUPER = SUPER_WORDS / WORDS_PER_THREAD;   // 4
    if (tid % THREADS_PER_SUPER == 0) super[tid / THREADS_PER_SUPER] = tpre;
    if (tid == 0) super[NSUPER] = (uint32_t)tot;
    __syncthreads();
    const uint32_t sbase = super[tid / THREADS_PER_SUPER];
    // indices come out sorted: walk the set bits in order; zero the global accumulators
    uint32_t run = tpre;
#pragma unroll 1
    for (int i = 0; i < WORDS_PER_THREAD; ++i) {
        const int word = tid * WORDS_PER_THREAD + i;
        if (word >= words) break;
        sub[word] = (uint16_t)(tpre + local[i] - sbase);
        unsigned long long m = bm[word];
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            c_indices[out + run] = (IDX)(wlo + (uint64_t)word * 64 + (uint64_t)b);
            if (!in_lds) c_data[out + run] = 0.0;
            ++run;
        }
    }
    __syncthreads();   // sub/super complete; zeroed accumulators have reached L2 (release at workgroup scope)
    mark(1);   // prefix + emit indices

    // Each wave OWNS a contiguous range of superblocks and walks every k in ascending order for
    // it: one owner per accumulator, reference order.  The ranges are balanced by PRODUCTS when
    // the column-bucket table of B is available (one bucket = one superblock, so the products of
    // superblock b are sum_k (t_k[b+1] - t_k[b]) — no pass over the entries needed), else by
    // outputs.
    const int nsuper = (words + SUPER_WORDS - 1) / SUPER_WORDS;
    __shared__ uint32_t pcnt[NSUPER];
    __shared__ uint32_t ppre[NSUPER + 1];
    if (B.bucket) {
        if (tid < NSUPER) pcnt[tid] = 0;
        __syncthreads();
        const uint64_t wb0 = wlo >> BUCKET_LOG2;
        const uint64_t i0 = wb0 + lane, i1 = wb0 + lane + 64, last = B.nb - 1;
        uint32_t c0 = 0, c1 = 0;
        for (uint64_t p = as + wave; p < ae; p += LG_WAVES) {
            const uint32_t *t = B.bucket + (uint64_t)A.indices[p] * B.nb;
            if ((int)lane < nsuper) c0 += t[i0 + 1 < last ? i0 + 1 : last] - t[i0 < last ? i0 : last];
            if ((int)lane + 64 < nsuper) c1 += t[i1 + 1 < last ? i1 + 1 : last] - t[i1 < last ? i1 : last];
        }
        if (c0) atomicAdd(&pcnt[lane], c0);
        if (c1) atomicAdd(&pcnt[lane + 64], c1);
        __syncthreads();
        uint64_t ptot;
        const uint64_t ex = block_excl_scan_u64((int)tid < nsuper ? pcnt[tid] : 0u, wt, &ptot);
        if ((int)tid < nsuper) ppre[tid] = (uint32_t)ex;
        if ((int)tid == nsuper) ppre[tid] = (uint32_t)ptot;
        __syncthreads();
    }
    mark(2);   // product counts per superblock
    // number of superblocks sb in [from, to) with bal[sb] - bal[from] < target
    auto boundary = [&](const uint32_t *bal, uint32_t from, uint32_t to, uint32_t target) -> uint32_t {
        uint32_t n = 0;
        const uint32_t b0 = bal[from];
        for (uint32_t sb = from + lane; sb < to; sb += WAVE) n += (bal[sb] - b0 < target) ? 1u : 0u;
        return __shfl((uint32_t)wave_sum_u64(n), 0, WAVE);
    };
    uint32_t sb_lo = 0, sb_hi = 0;
    if (!B.bucket) {
        sb_lo = boundary(super, 0, (uint32_t)nsuper, (uint32_t)(((uint64_t)cnt * wave) / LG_WAVES));
        sb_hi = boundary(super, 0, (uint32_t)nsuper, (uint32_t)(((uint64_t)cnt * (wave + 1)) / LG_WAVES));
        if (wave == 0) sb_lo = 0;
        if (wave == LG_WAVES - 1) sb_hi = (uint32_t)nsuper;
    }
    const uint64_t clo = wlo + (uint64_t)sb_lo * (SUPER_WORDS * 64);
    uint64_t chi = wlo + (uint64_t)sb_hi * (SUPER_WORDS * 64);
    if (chi > whi) chi = whi;

    uint32_t n_steps = 0, n_entries = 0;            // profiling only
    // apply one batch of up to 64 sub-ranges [s,e) (lane j holds k_j's), ascending j == ascending k
    auto apply_batch = [&](uint64_t s, uint64_t e, double av) {
        // PF steps (k's) are loaded together before any of them is applied: a step's entries
        // are a dependent global load (~1.4 us under load), and with 8 waves per CU one step at
        // a time left the CU 85 % idle.  Application order stays ascending k.
        constexpr int PF = 8;
        unsigned long long live = __ballot(e > s);
        if (prof) {
            n_steps += (uint32_t)__popcll(live);
            n_entries += (uint32_t)wave_sum_u64(e - s);
        }
        while (live) {
            int js[PF];
            int n = 0;
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                js[d] = 0;
                if (live) {
                    js[d] = __ffsll((long long)live) - 1;
                    live &= live - 1;
                    n = d + 1;
                }
            }
            uint64_t sj[PF], ej[PF];
            uint64_t cc[PF];
            double vv[PF];
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                sj[d] = readlane_u64(s, js[d]);
                ej[d] = readlane_u64(e, js[d]);
                cc[d] = 0;
                vv[d] = 0.0;
                if (d < n && sj[d] + lane < ej[d]) {
                    cc[d] = (uint64_t)B.indices[sj[d] + lane];
                    vv[d] = B.data[sj[d] + lane];
                }
            }
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                if (d < n) {                                  // wave-uniform
                    const double avj = readlane_f64(av, js[d]);
                    for (uint64_t b = sj[d] + lane; b < ej[d]; b += WAVE) {
                        uint64_t c;
                        double bv;
                        if (b == sj[d] + lane) {
                            c = cc[d];
                            bv = vv[d];
                        } else {                              // sub-range longer than one wave: rare
                            c = (uint64_t)B.indices[b];
                            bv = B.data[b];
                        }
                        c -= wlo;
                        const double pr = avj * bv;
                        const uint32_t word = (uint32_t)(c >> 6);
                        const uint32_t rank = super[word / SUPER_WORDS] + sub[word] +
                                              (uint32_t)__popcll(bm[word] & ((1ull << (c & 63)) - 1ull));
                        if (in_lds) {
                            acc[rank - base_rank] += pr;        // LDS, one owner wave, program order
                        } else {
                            double *dst = c_data + out + rank;
                            // The accumulator lives in this XCD's L2.  LOAD with sc1 (served by L2, around
                            // the per-CU L1 that other waves' stores never refresh); STORE plain: a plain
                            // store is written through to L2 and KEEPS the line there, whereas an sc1 /
                            // atomic store drops it to memory (profiles/r01q_spgemm_pmc.txt).
                            double v = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v += pr;
                            *(volatile double *)dst = v;
                        }
                    }
                    // the next k may hit the same accumulators: its loads must follow these stores
                    if (in_lds) __builtin_amdgcn_wave_barrier();
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
        }
    };

    if (B.bucket) {
        // PASSES: the window's superblocks are cut greedily into ranges [pb, pe) holding at most
        // ACC_CAP outputs (one superblock has <= 4096), so the accumulators of a pass always fit in
        // LDS.  (Before: tasks with more outputs accumulated through L2 with one global round trip
        // per k — 28 % of config 5's tasks took 46 % of this kernel's time, tests/spgemm_bench.py
        // with SPGEMM_PROF=1.)  Inside a pass the k metadata is gathered ONCE per workgroup into
        // LDS: row start, the bucket offsets at the 9 wave boundaries, the A value.
        __shared__ uint64_t k_row0[K_CAP];
        __shared__ double k_val[K_CAP];
        __shared__ uint32_t k_bnd[K_CAP][LG_WAVES + 1];
        __shared__ uint32_t w_bnd[LG_WAVES + 1];
        const uint64_t wb0 = wlo >> BUCKET_LOG2, last = B.nb - 1;
        uint32_t pb = 0;
        while (pb < (uint32_t)nsuper) {
            uint32_t pe = pb + 1;
            while (pe < (uint32_t)nsuper && super[pe + 1] - super[pb] <= (uint32_t)ACC_CAP) ++pe;
            base_rank = super[pb];
            const uint32_t pass_out = super[pe] - base_rank;
            if (pass_out) {                                  // block-uniform
                for (uint32_t i = tid; i < pass_out; i += LG_BLOCK) acc[i] = 0.0;
                // wave ownership inside the pass, balanced by products
                const uint32_t pprod = ppre[pe] - ppre[pb];
                const uint32_t lo_w = pb + boundary(ppre, pb, pe, (uint32_t)(((uint64_t)pprod * wave) / LG_WAVES));
                if (lane == 0) w_bnd[wave] = wave == 0 ? pb : lo_w;
                if (tid == 0) w_bnd[LG_WAVES] = pe;
                __syncthreads();
                const bool mine = w_bnd[wave + 1] > w_bnd[wave];
                for (uint64_t kc = as; kc < ae; kc += K_CAP) {
                    const uint32_t n = (ae - kc < (uint64_t)K_CAP) ? (uint32_t)(ae - kc) : (uint32_t)K_CAP;
                    for (uint32_t i = tid; i < n * (LG_WAVES + 1); i += LG_BLOCK) {
                        const uint32_t kk = i / (LG_WAVES + 1), g = i % (LG_WAVES + 1);
                        const uint64_t k = (uint64_t)A.indices[kc + kk];
                        const uint64_t bi = wb0 + w_bnd[g];
                        k_bnd[kk][g] = B.bucket[k * B.nb + (bi < last ? bi : last)];
                        if (g == 0) {
                            k_row0[kk] = (uint64_t)B.indptr[k];
                            k_val[kk] = A.data[kc + kk];
                        }
                    }
                    __syncthreads();
                    mark(8);
                    if (mine) {
                        for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
                            const uint32_t j = j0 + lane;
                            uint64_t s = 0, e = 0;
                            double av = 0.0;
                            if (j < n) {
                                s = k_row0[j] + k_bnd[j][wave];
                                e = k_row0[j] + k_bnd[j][wave + 1];
                                av = k_val[j];
                            }
                            apply_batch(s, e, av);
                        }
                    }
                    __syncthreads();
                    mark(9);
                }
                for (uint32_t i = tid; i < pass_out; i += LG_BLOCK) c_data[out + base_rank + i] = acc[i];
                __syncthreads();
                mark(10);
                if (prof && tid == 0) atomicAdd(&prof[11], 1ull);
            }
            pb = pe;
        }
        if (prof && lane == 0) {
            atomicAdd(&prof[12], (unsigned long long)n_steps);
            atomicAdd(&prof[13], (unsigned long long)n_entries);
        }
    } else if (sb_hi > sb_lo && clo < chi) {
        for (uint64_t p0 = as; p0 < ae; p0 += WAVE) {
            const uint64_t p = p0 + lane;
            const bool valid = p < ae;
            const uint64_t k = valid ? (uint64_t)A.indices[p] : 0;
            const double av = valid ? A.data[p] : 0.0;
            uint64_t s = valid ? (uint64_t)B.indptr[k] : 0, e = valid ? (uint64_t)B.indptr[k + 1] : 0;
            if (e > s) row_window(B, k, clo, chi, s, e);
            apply_batch(s, e, av);
        }
    }
    if (in_lds && !B.bucket) {
        __syncthreads();
        mark(3);   // accumulate (LDS accumulators)
        for (uint32_t i = tid; i < cnt; i += LG_BLOCK) c_data[out + i] = acc[i];
        mark(5);   // write back
    } else if (!in_lds) {
        mark(4);   // accumulate (L2 accumulators)
    }
    if (prof && tid == 0) atomicAdd(&prof[in_lds ? 6 : 7], 1ull);   // task counts
}

template <typename PTR>
__global__ void write_indptr_kernel(const uint64_t *__restrict__ first_task, const uint64_t *__restrict__ off,
                                    uint64_t rows, PTR *__restrict__ indptr) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    indptr[r] = (PTR)off[first_task[r]];
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T>
    T *as() { return (T *)p; }
};

template <typename IDX, typename PTR>
int32_t spgemm_impl(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c_out) {
    hipStream_t stream = nullptr;
    const uint64_t rows = a->rows, b_cols = b->cols;
    CsrView<IDX, PTR> A{(const PTR *)a->indptr, (const IDX *)a->indices, a->data, nullptr, 0};
    CsrView<IDX, PTR> B{(const PTR *)b->indptr, (const IDX *)b->indices, b->data, nullptr, 0};
    // column-bucket table of B (4 bytes per 4096 columns per row) when it fits a budget of 8 GiB
    DevBuf bucket;
    {
        const uint64_t nb = (b_cols >> BUCKET_LOG2) + 2;
        const uint64_t bytes = b->rows * nb * sizeof(uint32_t);
        if (options().spgemm_bucket && b->rows && bytes <= (8ull << 30)) {
            SPRS_TRY_HIP(bucket.alloc(bytes));
            uint64_t blocks = (b->rows + 3) / 4;
            if (blocks > 256 * 64) blocks = 256 * 64;
            hipLaunchKernelGGL((build_bucket_kernel<IDX, PTR>), dim3((unsigned)blocks), dim3(256), 0, stream, B.indptr,
                               B.indices, b->rows, nb, bucket.as<uint32_t>());
            SPRS_TRY_HIP(hipGetLastError());
            B.bucket = bucket.as<uint32_t>();
            B.nb = nb;
        }
    }

    DevBuf ub, ntasks, first_task, counters, wlog;
    SPRS_TRY_HIP(wlog.alloc(rows));
    SPRS_TRY_HIP(ub.alloc(rows * 8));
    SPRS_TRY_HIP(ntasks.alloc(rows * 8));
    SPRS_TRY_HIP(first_task.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(counters.alloc(16));
    SPRS_TRY_HIP(hipMemsetAsync(counters.p, 0, 16, stream));
    uint64_t ntask_total = 0;
    if (rows) {
        uint64_t blocks = (rows + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL((row_work_kernel<IDX, PTR>), dim3((unsigned)blocks), dim3(256), 0, stream, A, B, rows, b_cols,
                           (uint64_t)options().spgemm_heavy, ub.as<uint64_t>(), ntasks.as<uint64_t>(), wlog.as<uint8_t>());
        SPRS_TRY_HIP(hipGetLastError());
    }
    SPRS_TRY(exclusive_scan_u64(ntasks.as<uint64_t>(), first_task.as<uint64_t>(), rows, stream));
    SPRS_TRY_HIP(hipMemcpy(&ntask_total, first_task.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost));

    DevBuf task_row, small_list, large_list, count, off;
    SPRS_TRY_HIP(task_row.alloc(ntask_total * 8));
    SPRS_TRY_HIP(small_list.alloc(ntask_total * 8));
    SPRS_TRY_HIP(large_list.alloc(ntask_total * 8));
    SPRS_TRY_HIP(count.alloc(ntask_total * 8));
    SPRS_TRY_HIP(off.alloc((ntask_total + 1) * 8));
    uint64_t n_small = 0, n_large = 0;
    if (ntask_total) {
        hipLaunchKernelGGL(make_tasks_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream,
                           ub.as<uint64_t>(), ntasks.as<uint64_t>(), first_task.as<uint64_t>(), rows,
                           task_row.as<uint64_t>(), small_list.as<uint64_t>(), large_list.as<uint64_t>(),
                           counters.as<unsigned long long>());
        SPRS_TRY_HIP(hipGetLastError());
        uint64_t h[2];
        SPRS_TRY_HIP(hipMemcpy(h, counters.p, 16, hipMemcpyDeviceToHost));
        n_small = h[0];
        n_large = h[1];
    }
    auto small_grid = [&]() {
        uint64_t g = (n_small + SM_WAVES - 1) / SM_WAVES;
        if (g > 256 * 32) g = 256 * 32;
        return dim3((unsigned)g);
    };
    if (n_large > 0x7fffffffull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "too many SpGEMM tasks for one launch");

    // ---- symbolic ----------------------------------------------------------
    if (n_small) {
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, false>), small_grid(), dim3(SM_BLOCK), 0, stream, A, B,
                           small_list.as<uint64_t>(), n_small, task_row.as<uint64_t>(), ub.as<uint64_t>(),
                           count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (n_large) {
        hipLaunchKernelGGL((large_symbolic_kernel<IDX, PTR>), dim3((unsigned)n_large), dim3(LG_BLOCK), 0, stream, A, B,
                           b_cols, large_list.as<uint64_t>(), task_row.as<uint64_t>(), first_task.as<uint64_t>(),
                           ntasks.as<uint64_t>(), wlog.as<uint8_t>(), count.as<uint64_t>());
        SPRS_TRY_HIP(hipGetLastError());
    }

    // ---- prefix sum of the counts -> offsets, C.indptr (smmp.rs:320-331) ----
    SPRS_TRY(exclusive_scan_u64(count.as<uint64_t>(), off.as<uint64_t>(), ntask_total, stream));
    uint64_t c_nnz = 0;
    SPRS_TRY_HIP(hipMemcpy(&c_nnz, off.as<uint64_t>() + ntask_total, 8, hipMemcpyDeviceToHost));
    if (sizeof(PTR) == 4 && c_nnz > 0xFFFFFFFFull)
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold the nnz of the product (%llu)",
                  (unsigned long long)c_nnz);   // Iptr::from_usize, smmp.rs:121

    sprs_hip_csmat *c = nullptr;
    SPRS_TRY(alloc_csmat(&c, SPRS_HIP_CSR, rows, b_cols, c_nnz, (int32_t)sizeof(PTR), (int32_t)sizeof(IDX)));
    hipLaunchKernelGGL((write_indptr_kernel<PTR>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream,
                       first_task.as<uint64_t>(), off.as<uint64_t>(), rows, (PTR *)c->indptr);

    // ---- numeric -------------------------------------------------------------
    DevBuf prof;
    if (options().spgemm_prof) {
        SPRS_TRY_HIP(prof.alloc(128));
        SPRS_TRY_HIP(hipMemsetAsync(prof.p, 0, 128, stream));
    }
    if (n_small)
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, true>), small_grid(), dim3(SM_BLOCK), 0, stream, A, B,
                           small_list.as<uint64_t>(), n_small, task_row.as<uint64_t>(), ub.as<uint64_t>(),
                           count.as<uint64_t>(), off.as<uint64_t>(), (IDX *)c->indices, c->data);
    if (n_large)
        hipLaunchKernelGGL((large_numeric_kernel<IDX, PTR>), dim3((unsigned)n_large), dim3(LG_BLOCK), 0, stream, A, B,
                           b_cols, large_list.as<uint64_t>(), task_row.as<uint64_t>(), first_task.as<uint64_t>(),
                           ntasks.as<uint64_t>(), wlog.as<uint8_t>(), count.as<uint64_t>(), off.as<uint64_t>(),
                           (IDX *)c->indices, c->data, prof.as<unsigned long long>());
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        sprs_hip_csmat_free(c);
        return fail_hip(e, "spgemm numeric");
    }
    if (prof.p) {
        unsigned long long h[16];
        if (hipMemcpy(h, prof.p, 128, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr,
                    "[spgemm_prof] inside accumulate: staging %llu, apply %llu, writeback %llu cycles; passes %llu, "
                    "k-steps %llu (all waves), entries %llu\n",
                    h[8], h[9], h[10], h[11], h[12], h[13]);
            fprintf(stderr,
                    "[spgemm_prof] thread-0 cycles summed over large tasks: bits %llu, prefix+emit %llu, prodcount %llu, "
                    "accumulate(LDS) %llu, accumulate(L2) %llu, writeback %llu; tasks LDS %llu, L2 %llu\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        }
    }
    *c_out = c;
    return SPRS_HIP_OK;
}

}  // namespace

int32_t spgemm_f64(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c) {
    if (b->cols > 0xFFFFFFFEull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "SpGEMM: more than 2^32-2 columns is not supported");
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return spgemm_impl<uint64_t, uint64_t>(a, b, c);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return spgemm_impl<uint32_t, uint64_t>(a, b, c);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return spgemm_impl<uint64_t, uint32_t>(a, b, c);
    return spgemm_impl<uint32_t, uint32_t>(a, b, c);
}

}  // namespace sprs_hip
