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
//     task handled by ONE WAVE with an LDS hash table (keys + f64 accumulators): the
//     products of the row are walked 64 at a time in the reference's order, keys inserted in
//     parallel, products added through order tags; then a bitonic sort of the table in LDS.
//     Rows of <= 64 products use 128-slot tables (32 waves per CU).
//   * larger rows get one task per column window — 2^17 columns by default, narrowed down to
//     2^13 for heavy rows so that a hub row becomes many tasks — handled by a 512-thread
//     workgroup with an LDS BITMAP of the window: setting bits is the symbolic pass, a popcount
//     prefix over the bitmap turns a column into its rank inside the (sorted!) output row, so
//     indices come out sorted for free.  Values are accumulated in LDS, in passes of a few
//     thousand outputs (superblock ranges of the window); within a pass the expansion is walked
//     entry-parallel, 2048 products per chunk, and products that meet in one accumulator are
//     added in position order through LDS order tags (see large_numeric_kernel).
//   * a column-bucket table of B (entries of every row before each 2048-column boundary,
//     built per call, 4 B per row per 2048 columns) replaces the binary searches that
//     locate a row inside a window / superblock range.
// Per-task counts are scanned (hand-written two-level prefix sum, scan.hip) into output
// offsets; C.indptr falls out of the same scan.  Integer/LDS/HBM-bound: no MFMA.
// History of what was measured (profiles/): per-row windows + LDS accumulators
// 1.96 s -> 0.50 s on config 5; bucket table, LDS staging, prefetch -> 0.454 s (waves owning
// column ranges, one k at a time); entry-parallel expansion with order tags -> 0.22 s.
// SPGEMM_PROF (option spgemm_prof) prints the phase profile of the large-row numeric kernel.
#include "common.hpp"
#include "scan.hpp"

#include <vector>

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint64_t SMALL_MAX = 512;       // products per row handled by the wave/hash path
constexpr int SMALL_TAB = 1024;           // hash slots per wave (load factor <= 0.5)
constexpr uint64_t TINY_MAX = 64;         // rows of at most this many products: same kernel with a 128-slot table,
constexpr int TINY_TAB = 128;             //   so that 32 waves share a CU instead of 12 (these rows are latency bound)
constexpr int SM_BLOCK = 256;             // 4 waves
constexpr int SM_WAVES = SM_BLOCK / WAVE;
constexpr int MAX_WIN_LOG2 = 19;          // widest column window of a large-row task (option spgemm_winlog <= this)
constexpr int LG_BLOCK = 512;             // 8 waves
constexpr int LG_WAVES = LG_BLOCK / WAVE;
constexpr int SUPER_WORDS = 32;           // bitmap words per superblock (2048 columns)

// LDS layout of the large-row kernels for windows of up to 2^WL columns.  The narrower the window
// the more workgroups share a CU (each phase of a task ends in a barrier, so a lone workgroup
// leaves the CU idle while its loads are in flight): 2^19 -> 1 per CU, 2^18 -> 2, 2^17 -> 3, 2^16 -> 4.
// Measured on config 5: 257 / 243 / 185 / 300 ms for the numeric kernel -> the default is 2^17.
template <int WL>
struct LgCfg {
    static constexpr int WORDS = 1 << (WL - 6);                 // 64-bit bitmap words
    static constexpr int WPT = WORDS / LG_BLOCK;                // words per thread in the popcount prefix
    static constexpr int NSUPER = WORDS / SUPER_WORDS;
    // accumulators of one pass; a superblock alone (<= 2048 outputs) must fit
    static constexpr int ACC_CAP = WL >= 19 ? 6144 : WL == 18 ? 3072 : WL == 17 ? 3072 : 2048;
    static constexpr int K_CAP = WL >= 18 ? 512 : 256;          // k's staged in LDS per group (<= one per thread)
    static_assert(WL >= 16 && WL <= MAX_WIN_LOG2, "window");
    static_assert(ACC_CAP >= SUPER_WORDS * 64, "a superblock must fit one pass");
};

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
    // row k with column < b * 2048.  Turns "where does row k enter column window [lo,hi)" — two
    // binary searches, ~20 dependent loads per (row, window) — into two independent loads.
    const uint32_t *bucket;
    uint64_t nb;
};

constexpr int BUCKET_LOG2 = 11;   // 2048 columns = one superblock of the LDS bitmap: pass and window bounds are free

// sub-range [s,e) of row k (given its [s,e) = whole row) inside columns [lo, hi).  With the bucket
// table a bound that is a multiple of 2048 costs one load (callers round hi UP past the last
// column instead of clamping it); any other bound adds a binary search inside its bucket.
template <typename IDX, typename PTR>
__device__ __forceinline__ void row_window(const CsrView<IDX, PTR> &B, uint64_t k, uint64_t lo, uint64_t hi,
                                           uint64_t &s, uint64_t &e) {
    if (B.bucket) {
        const uint32_t *t = B.bucket + k * B.nb;
        const uint64_t row0 = s;
        auto first_ge = [&](uint64_t v) -> uint64_t {
            const uint64_t b = v >> BUCKET_LOG2;
            if (b >= B.nb - 1) return row0 + t[B.nb - 1];          // past the last column: the whole row
            const uint64_t p0 = row0 + t[b];
            if ((v & ((1ull << BUCKET_LOG2) - 1)) == 0) return p0;
            return lower_bound_col(B.indices, p0, row0 + t[b + 1], v);
        };
        s = first_ge(lo);
        e = first_ge(hi);
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

__device__ __forceinline__ uint32_t hash_slot(uint32_t c, int lg) { return (c * 0x9E3779B1u) >> (32 - lg); }

__device__ __forceinline__ int ceil_log2_u32(uint32_t v) { return v <= 1 ? 0 : 32 - __clz(v - 1); }

// ---------------------------------------------------------------------------
// pass 0: per-row product count and number of tasks
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void row_work_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B, uint64_t rows,
                                                       uint64_t b_cols, uint64_t heavy_products, uint32_t wl,
                                                       uint64_t *__restrict__ ub, uint64_t *__restrict__ ntasks) {
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
            // A large row is ONE task that walks its column windows (2^wl columns each) one after the other; only a
            // heavy row (hub) is cut into several tasks of consecutive windows, ~heavy_products products each, so that
            // it does not become one serial chain at the end of the launch.  large_rows_kernel derives the same
            // windows-per-task from ntasks[r].
            uint64_t nt = acc ? 1 : 0;
            if (acc > SMALL_MAX) {
                uint64_t nwin = (b_cols + (1ull << wl) - 1) >> wl;
                if (nwin == 0) nwin = 1;
                uint64_t want = acc / heavy_products;
                if (want < 1) want = 1;
                if (want > nwin) want = nwin;
                const uint64_t wpt = (nwin + want - 1) / want;
                nt = (nwin + wpt - 1) / wpt;
            }
            ntasks[r] = nt;
        }
    }
}

// Task lists, deterministic (first version: atomicAdd tickets, i.e. an arbitrary order that changed from call to call).
// Classes of a row: tiny (<= 64 products), small (<= 512), large (one task per column window).
__global__ void task_class_kernel(const uint64_t *__restrict__ ub, const uint64_t *__restrict__ ntasks, uint64_t rows,
                                  uint64_t *__restrict__ is_tiny, uint64_t *__restrict__ is_small, uint64_t *__restrict__ n_large) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t n = ntasks[r], u = ub[r];
    is_tiny[r] = (n && u <= TINY_MAX) ? 1 : 0;
    is_small[r] = (n && u > TINY_MAX && u <= SMALL_MAX) ? 1 : 0;
    n_large[r] = (n && u > SMALL_MAX) ? n : 0;
}

// lists in row order; for the large tasks also the sort key: the cost class (log2 of the products per task), costliest
// first, so that the long tasks start early and the short ones fill the tail of the launch (the sort is stable: inside
// a class the tasks stay in row order, and the list — with it every launch — is the same run to run)
__global__ void make_tasks_kernel(const uint64_t *__restrict__ ntasks, const uint64_t *__restrict__ first_task,
                                  const uint64_t *__restrict__ ub, uint64_t rows, const uint64_t *__restrict__ pos_tiny,
                                  const uint64_t *__restrict__ pos_small, const uint64_t *__restrict__ pos_large,
                                  const uint64_t *__restrict__ is_tiny, const uint64_t *__restrict__ is_small,
                                  uint64_t *__restrict__ task_row, uint64_t *__restrict__ tiny_list,
                                  uint64_t *__restrict__ small_list, uint64_t *__restrict__ large_list,
                                  uint64_t *__restrict__ large_key) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t n = ntasks[r];
    if (!n) return;
    const uint64_t f = first_task[r];
    for (uint64_t j = 0; j < n; ++j) task_row[f + j] = r;
    if (is_tiny[r]) {
        tiny_list[pos_tiny[r]] = f;
    } else if (is_small[r]) {
        small_list[pos_small[r]] = f;
    } else {
        const uint64_t pos = pos_large[r];
        const uint64_t cost = ub[r] / n;
        const uint64_t key = (uint64_t)__clzll((long long)(cost | 1));     // 0 .. 63, small = costly
        for (uint64_t j = 0; j < n; ++j) {
            large_list[pos + j] = f + j;
            large_key[pos + j] = key;
        }
    }
}

// Block b runs on XCD b % 8 (observed; only speed depends on it).  Default: the task list is dealt round-robin (the
// tasks are sorted by cost, so every XCD gets the same mix).  spgemm_xcd_chunk = -1 gives every XCD one contiguous run
// instead (tasks sharing a column window of B stay in one L2) — measured slower on config 5: the runs differ in cost.
__device__ __forceinline__ uint64_t task_of_block(uint64_t bid, uint64_t n, uint32_t chunk) {
    if (chunk == 0) return bid;                                  // round-robin over the XCDs
    if (chunk == 0xFFFFFFFFu) {                                  // one contiguous run per XCD (a bijection on [0, n))
        const uint64_t q = n >> 3, rem = n & 7, k = bid & 7, j = bid >> 3;
        return k * q + (k < rem ? k : rem) + j;
    }
    return bid;
}

// ---------------------------------------------------------------------------
// small rows: one wave per task, LDS hash table
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR, bool NUMERIC, int TAB>
__global__ __launch_bounds__(SM_BLOCK) void small_rows_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B,
                                                              const uint64_t *__restrict__ small_list,
                                                              uint64_t n_small, const uint64_t *__restrict__ task_row,
                                                              const uint64_t *__restrict__ ub,
                                                              uint64_t *__restrict__ count,        // symbolic: out
                                                              const uint64_t *__restrict__ off,    // numeric: in
                                                              IDX *__restrict__ c_indices, double *__restrict__ c_data) {
    __shared__ uint32_t keys_s[SM_WAVES][TAB];
    __shared__ double vals_s[NUMERIC ? SM_WAVES : 1][NUMERIC ? TAB : 1];
    __shared__ uint32_t tag_s[NUMERIC ? SM_WAVES : 1][WAVE];   // order tags of the entry-parallel path
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
        wave_sync_lds();
        const uint64_t as = (uint64_t)A.indptr[r], ae = (uint64_t)A.indptr[r + 1];
        uint32_t fresh = 0;
        if (ae - as <= (uint64_t)WAVE) {
            // ENTRY-PARALLEL path (rows with at most 64 k's — nearly all small rows): lane j holds k_j; the
            // concatenation of the B rows (k ascending: the reference's order) is walked 64 products at a time,
            // every load independent.  (One k at a time — below — leaves 59 of 64 lanes idle on a banded matrix
            // and chains one memory round trip per k: 5-pt Laplacian squared, 51 ms of 71.)
            const bool has = as + lane < ae;
            const uint64_t k = has ? (uint64_t)A.indices[as + lane] : 0;
            const double av = has ? A.data[as + lane] : 0.0;
            const uint64_t bs = has ? (uint64_t)B.indptr[k] : 0;
            const uint32_t len = has ? (uint32_t)((uint64_t)B.indptr[k + 1] - bs) : 0u;
            uint32_t inc = len;                               // inclusive prefix of the lengths over the lanes
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, WAVE);
                if (lane >= (uint32_t)off) inc += o;
            }
            const uint32_t total = __shfl(inc, WAVE - 1, WAVE);
            if constexpr (NUMERIC) tag_s[wave][lane] = EMPTY;
            for (uint32_t base = 0; base < total; base += WAVE) {
                const uint32_t t = base + lane;
                const bool valid = t < total;
                uint32_t own = 0;                             // owner = number of lanes whose prefix is <= t
#pragma unroll
                for (int step = WAVE / 2; step > 0; step >>= 1) {
                    const uint32_t v = __shfl(inc, (int)(own + step - 1), WAVE);
                    if (v <= t) own += step;
                }
                own &= WAVE - 1;
                const uint32_t inc_o = __shfl(inc, (int)own, WAVE), len_o = __shfl(len, (int)own, WAVE);
                const uint64_t bs_o = __shfl(bs, (int)own, WAVE);
                const double av_o = __shfl(av, (int)own, WAVE);
                const uint64_t pos = bs_o + (uint64_t)(t - (inc_o - len_o));
                const uint32_t c = valid ? (uint32_t)B.indices[pos] : 0u;
                double pr = 0.0;
                if constexpr (NUMERIC) pr = valid ? av_o * B.data[pos] : 0.0;
                uint32_t h = hash_slot(c, lg);
                if (valid) {
                    for (;;) {
                        const uint32_t old = atomicCAS(&keys[h], EMPTY, c);
                        if (old == EMPTY) {
                            ++fresh;
                            if constexpr (NUMERIC) vals[h] = 0.0;       // tmp starts at N::zero()
                            break;
                        }
                        if (old == c) break;
                        h = (h + 1) & mask;
                    }
                }
                if constexpr (NUMERIC) {
                    // Two lanes of this batch may hold the same column (from different k's): their products must be
                    // added in lane order = k order.  64 direct-mapped order tags: the lowest pending lane of a tag
                    // adds, the others (same column, or merely the same tag) take another turn.
                    wave_sync_lds();
                    bool pend = valid;
                    uint32_t *tg = &tag_s[wave][h & (WAVE - 1)];
                    while (__ballot(pend)) {
                        if (pend) atomicMin(tg, lane);
                        wave_sync_lds();
                        if (pend && *(volatile uint32_t *)tg == lane) {
                            vals[h] += pr;
                            *(volatile uint32_t *)tg = EMPTY;
                            pend = false;
                        }
                        wave_sync_lds();
                    }
                }
            }
        } else
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
                wave_sync_lds();
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
                    wave_sync_lds();
                }
            }
            const uint64_t o = off[t];
            for (uint32_t i = lane; i < need; i += WAVE) {
                if (c_indices) c_indices[o + i] = (IDX)keys[i];   // null: C already has its structure (numeric on a kept plan)
                if (c_data) c_data[o + i] = vals[i];      // null: structure only (the twin of smmp::symbolic)
            }
            wave_sync_lds();
        }
    }
}

// ---------------------------------------------------------------------------
// large rows: one workgroup per task = (row i, a run of consecutive column windows of 2^WL columns)
//
// Round 1 / early round 2 made every (row, window) pair its own workgroup: 2.3 M tasks of 2 300 products on config 5,
// each a chain of ~8 dependent memory round trips (task -> row -> A_i -> B.indptr / bucket table -> entries -> ...) plus a
// dozen workgroup barriers — the kernels were bound by that fixed cost per task (~50 of 68 us), not by the products
// (profiles/r02s: symbolic 46 ms, numeric 203 ms; window width 2^19, i.e. 4x fewer tasks at a third of the occupancy,
// measured the same).  Now a task is a ROW (hub rows: a few runs of windows, option spgemm_heavy), its windows are
// walked one after the other by the same workgroup, and for the rows whose k's fit one staged group (<= K_CAP; all but
// ~6 000 rows of config 5) thread j keeps k_j, the bounds of B's row k_j and a_ik in REGISTERS for the whole task:
// per window it needs ONE bucket-table load (where row k_j crosses the next window edge), issued one window ahead.
//
// Per window:  (1) BIT PASS — the concatenation of the k's sub-ranges (the task's expansion in the reference's own
// order: k ascending, columns ascending inside a k) is walked in batches of 64 consecutive positions per wave
// instruction (coalesced loads of B's column ids), every entry sets its bit in an LDS bitmap of the window (ds_or);
// (2) a popcount prefix over the bitmap (super[] per 2048 columns + 16-bit sub[] per word) turns a column into its rank
// in the output row — the indices come out SORTED without a sort; the symbolic kernel stops at the popcount;
// (3) VALUES — the window's superblocks are cut greedily into passes of <= ACC_CAP outputs whose accumulators live in
// LDS; the expansion restricted to the pass is walked again in the same batches.
//
// ORDER OF THE ADDITIONS.  The reference builds C(i,j) by a fixed chain (k ascending from +0.0, smmp.rs:174-181), so
// products that meet in one accumulator must be added in position order, bit for bit.  Earlier versions settled that
// with per-accumulator order tags and rounds (8.7 rounds and ~11 workgroup barriers per 2048 products).  Now the
// hardware's own ordering does it: the LDS executes the instructions of ONE wave in issue order, and the lanes of one
// ds_add_f64 that belong to the same k hit distinct accumulators (columns of a B row are distinct).  So
//   * a wave adds the products of its batch with one ds_add_f64 per k-run (a run = the lanes of one k, contiguous in
//     lane order; runs in ascending order) — fire and forget, no read-back, no tags;
//   * batches are dealt to the waves round-robin and a TOKEN in LDS (the index of the batch whose turn it is) is handed
//     from wave to wave: a wave loads its entries, computes ranks and products at its own pace, waits for its turn,
//     issues its adds, waits for them to complete (s_waitcnt lgkmcnt(0)) and passes the token on.  Only the adds are
//     serialised (a few instructions per batch); loads, searches and rank computations of all waves overlap.
// No float atomics race anywhere (every accumulator sees its additions in the reference's order) => values bit-exact
// and deterministic.  tests/test_spgemm_gpu.py compares bits with the oracle; the CPU emulator (tests/emu) runs the
// hand-over with the waves scheduled in reversed / rotated order.
// ---------------------------------------------------------------------------

// position of the first entry of row k with column >= v; the row's entries are [row0, ..), those before `lo` are known
// to be < v and those from `hi` on >= v.  With the bucket table a bound that is a multiple of 2048 costs one load
// (callers round an upper bound UP past the last column instead of clamping it).
template <typename IDX, typename PTR>
__device__ __forceinline__ uint64_t first_ge(const CsrView<IDX, PTR> &B, uint64_t k, uint64_t row0, uint64_t lo, uint64_t hi,
                                             uint64_t v) {
    if (B.bucket) {
        const uint32_t *t = B.bucket + k * B.nb;
        const uint64_t b = v >> BUCKET_LOG2;
        if (b >= B.nb - 1) return row0 + t[B.nb - 1];               // past the last column: the whole row
        const uint64_t p0 = row0 + t[b];
        if ((v & ((1ull << BUCKET_LOG2) - 1)) == 0) return p0;
        return lower_bound_col(B.indices, p0, row0 + t[b + 1], v);
    }
    return lower_bound_col(B.indices, lo, hi, v);
}

// Only the k's that HAVE entries in the range are kept (most do not, in a narrow window): the walks then cross exactly
// one boundary per k instead of idling through runs of empty k's.  One scan carries both the count of kept k's (high
// word) and the prefix of the lengths (low word).  kP[kept] = total, sentinels behind it for the search.
template <int K_CAP>
__device__ __forceinline__ uint32_t stage_compact(uint32_t len, uint64_t s, double av, uint64_t *kS, uint32_t *kP, double *kA,
                                                  uint64_t *wt) {
    const uint32_t tid = threadIdx.x;
    uint64_t tot;
    const uint64_t ex = block_excl_scan_u64_lds(((uint64_t)(len ? 1u : 0u) << 32) | len, wt, &tot);
    const uint32_t kept = (uint32_t)(tot >> 32), total = (uint32_t)tot;
    if (len) {
        const uint32_t j = (uint32_t)(ex >> 32);
        kS[j] = s;
        kP[j] = (uint32_t)ex;
        if (kA) kA[j] = av;
    }
    if (tid == 0) kP[kept] = total;
    for (uint32_t i = kept + 1 + tid; i <= (uint32_t)K_CAP; i += LG_BLOCK) kP[i] = 0xFFFFFFFFu;
    lds_barrier();
    return total;
}

// general form: the group's k's, their rows of B and the bounds are loaded here (rows with more than K_CAP k's)
template <int K_CAP, typename IDX, typename PTR>
__device__ __forceinline__ uint32_t stage_k_group(const CsrView<IDX, PTR> &A, const CsrView<IDX, PTR> &B,
                                                  uint64_t kc, uint32_t n, uint64_t wlo, uint64_t whi, bool whole_row,
                                                  uint64_t *kS, uint32_t *kP, double *kA, uint64_t *wt) {
    const uint32_t tid = threadIdx.x;
    uint32_t len = 0;                      // < 2^32: a window holds at most 2^19 columns of a row
    uint64_t s = 0;
    double av = 0.0;
    if (tid < n) {
        const uint64_t k = (uint64_t)A.indices[kc + tid];
        uint64_t e = (uint64_t)B.indptr[k + 1];
        s = (uint64_t)B.indptr[k];
        // with the bucket table the window bounds do not depend on the row bounds: the two pairs of
        // loads go out together (one memory round trip less on the critical path)
        if (!whole_row && (B.bucket || e > s)) row_window(B, k, wlo, whi, s, e);
        len = (uint32_t)(e - s);
        if (kA) av = A.data[kc + tid];
    }
    return stage_compact<K_CAP>(len, s, av, kS, kP, kA, wt);
}

// owner of flat position t < total: the last k with kP[k] <= t.  kP[0 .. K_CAP] is non-decreasing
// (total at [kept], sentinels behind it), so a fixed-trip, branch-free descent finds it.
template <int K_CAP>
__device__ __forceinline__ uint32_t flat_owner(const uint32_t *kP, uint32_t t) {
    uint32_t lo = 0;
#pragma unroll
    for (int step = K_CAP / 2; step > 0; step >>= 1)
        if (kP[lo + step] <= t) lo += step;
    return lo;
}

// A lane's positions inside a batch are 64 apart: it searches its owner once and again only when a position has left
// the owner's run (long runs — the common case where the products are — cost one LDS compare per entry).
struct FlatWalk {
    uint32_t o = 0, nxt = 0;  // nxt = kP[o + 1]; 0 = nothing found yet
    uint64_t base = 0;        // kS[o] - kP[o]: position t lives at B entry base + t
    template <int K_CAP>
    __device__ __forceinline__ void seek(const uint64_t *kS, const uint32_t *kP, uint32_t t) {
        if (t < nxt) return;
        o = flat_owner<K_CAP>(kP, t);
        nxt = kP[o + 1];
        base = kS[o] - kP[o];
    }
};

constexpr int LG_U = 4;                    // wave instructions (of 64 consecutive positions) per batch, at most

// batch geometry of a walk over `gtot` positions: a batch is U wave instructions of 64 consecutive positions; few positions ->
// smaller batches, so that all waves get some.  Up to 64 LG_U LG_WAVES positions every wave has at most ONE batch.
__device__ __forceinline__ uint32_t batch_u(uint32_t gtot) {
    return gtot > 64u * 2 * LG_WAVES ? (uint32_t)LG_U : gtot > 64u * LG_WAVES ? 2u : 1u;
}

// the entries of one batch as a wave holds them: lane l, instruction u = position (b U + u) 64 + l
struct Batch {
    uint32_t cc[LG_U];        // column - first column of the window
    uint32_t own[LG_U];       // staged k the entry belongs to (ascending with the position)
    double pr[LG_U];          // a_ik * b_kj
    bool val[LG_U];
};

template <int K_CAP, bool VALUES, typename IDX>
__device__ __forceinline__ void batch_load(Batch &bt, const IDX *__restrict__ b_indices, const double *__restrict__ b_data,
                                           uint64_t wlo, uint32_t gtot, uint32_t U, uint32_t b, const uint64_t *kS,
                                           const uint32_t *kP, const double *kA) {
    const uint32_t t0 = b * (64 * U) + (threadIdx.x & (WAVE - 1));
    FlatWalk wk;
    uint64_t pos[LG_U];
    double av[LG_U];
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        const uint32_t t = t0 + 64u * u;
        bt.val[u] = (uint32_t)u < U && t < gtot;
        pos[u] = 0;
        bt.own[u] = 0;
        av[u] = 0.0;
        if (bt.val[u]) {
            wk.seek<K_CAP>(kS, kP, t);
            pos[u] = wk.base + t;
            bt.own[u] = wk.o;
            if constexpr (VALUES) av[u] = kA[wk.o];
        }
    }
    double bv[LG_U];
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {           // all loads of the batch are independent
        bt.cc[u] = bt.val[u] ? (uint32_t)((uint64_t)b_indices[pos[u]] - wlo) : 0u;
        bv[u] = 0.0;
        if constexpr (VALUES) bv[u] = bt.val[u] ? b_data[pos[u]] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < LG_U; ++u) bt.pr[u] = av[u] * bv[u];
}

__device__ __forceinline__ void batch_bits(const Batch &bt, uint32_t *bm32) {
#pragma unroll
    for (int u = 0; u < LG_U; ++u)
        if (bt.val[u]) atomicOr(&bm32[bt.cc[u] >> 5], 1u << (bt.cc[u] & 31));   // little endian: bit c & 63 of word c >> 6
}

template <int K_CAP, typename IDX>
__device__ __forceinline__ void walk_bits(const IDX *__restrict__ b_indices, uint64_t wlo, uint32_t gtot, const uint64_t *kS,
                                          const uint32_t *kP, uint32_t *bm32) {
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t U = batch_u(gtot);
    const uint32_t nbatch = (gtot + 64 * U - 1) / (64 * U);
    for (uint32_t b = wave; b < nbatch; b += LG_WAVES) {
        Batch bt;
        batch_load<K_CAP, false>(bt, b_indices, (const double *)nullptr, wlo, gtot, U, b, kS, kP, (const double *)nullptr);
        batch_bits(bt, bm32);
    }
}

__device__ __forceinline__ void token_wait(uint32_t *token, uint32_t turn) {
    while (*(volatile uint32_t *)token != turn) SPRS_POLL_PAUSE();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void token_pass(uint32_t *token, uint32_t next) {
    SPRS_LDS_FENCE();                      // this wave's adds have been performed
    if ((threadIdx.x & (WAVE - 1)) == 0) *(volatile uint32_t *)token = next;
}

// the products of one wave instruction (64 consecutive positions, owners ascending with the lane): one add instruction
// per k-run, runs in ascending order
__device__ __forceinline__ void add_runs(bool val, uint32_t own, uint32_t slot, double pr, double *acc, bool lds_atomic) {
    unsigned long long todo = __ballot(val);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t oo = (uint32_t)__builtin_amdgcn_readlane((int)own, first);
        const bool mine = val && own == oo;
        if (mine) {
            if (lds_atomic) {
                atomicAdd(&acc[slot], pr);               // ds_add_f64, no return value
            } else {
                volatile double *a = acc + slot;         // A/B switch: read, add, write (same order, three instructions)
                *a = *a + pr;
            }
        }
        todo &= ~__ballot(mine);
    }
}

// ranks of the batch's columns (before the turn: only the adds are serialised), then the adds when the token arrives
__device__ __forceinline__ void batch_add(const Batch &bt, uint32_t U, const unsigned long long *bm, const uint16_t *sub,
                                          const uint32_t *super, uint32_t base_rank, double *acc, uint32_t *token, uint32_t turn,
                                          bool lds_atomic) {
    uint32_t slot[LG_U];
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        const uint32_t word = bt.cc[u] >> 6;
        slot[u] = bt.val[u] ? super[word / SUPER_WORDS] + sub[word] +
                                  (uint32_t)__popcll(bm[word] & ((1ull << (bt.cc[u] & 63)) - 1ull)) - base_rank
                            : 0u;
    }
    token_wait(token, turn);
#pragma unroll
    for (int u = 0; u < LG_U; ++u)
        if ((uint32_t)u < U) add_runs(bt.val[u], bt.own[u], slot[u], bt.pr[u], acc, lds_atomic);
    token_pass(token, turn + 1);
}

// value walk of one staged group restricted to a pass; returns the number of batches (the token advances by it)
template <int K_CAP, typename IDX>
__device__ __forceinline__ uint32_t walk_values(const IDX *__restrict__ b_indices, const double *__restrict__ b_data, uint64_t wlo,
                                                uint32_t gtot, const uint64_t *kS, const uint32_t *kP, const double *kA,
                                                const unsigned long long *bm, const uint16_t *sub, const uint32_t *super,
                                                uint32_t base_rank, double *acc, uint32_t *token, uint32_t tok_base,
                                                bool lds_atomic) {
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t U = batch_u(gtot);
    const uint32_t nbatch = (gtot + 64 * U - 1) / (64 * U);
    for (uint32_t b = wave; b < nbatch; b += LG_WAVES) {
        Batch bt;
        batch_load<K_CAP, true>(bt, b_indices, b_data, wlo, gtot, U, b, kS, kP, kA);
        batch_add(bt, U, bm, sub, super, base_rank, acc, token, tok_base + b, lds_atomic);
    }
    return nbatch;
}

// OCC = waves per SIMD the register allocation aims at: 6 = three workgroups per CU (what the LDS of the 2^16 / 2^17
// layouts allows; 80 VGPRs, a few spills), 4 = two per CU with 128 VGPRs (option spgemm_occupancy, A/B)
template <int WL, typename IDX, typename PTR, bool NUMERIC, int OCC>
__global__ __launch_bounds__(LG_BLOCK, OCC) void large_rows_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B, uint64_t b_cols,
                                                              const uint64_t *__restrict__ large_list,
                                                              const uint64_t *__restrict__ task_row,
                                                              const uint64_t *__restrict__ first_task,
                                                              const uint64_t *__restrict__ ntasks,
                                                              uint64_t *__restrict__ count,       // symbolic: out
                                                              const uint64_t *__restrict__ off,   // numeric: in
                                                              IDX *__restrict__ c_indices, double *__restrict__ c_data,
                                                              uint32_t xcd_chunk, uint32_t flags) {
    using Cfg = LgCfg<WL>;
    constexpr int WORDS = Cfg::WORDS, WPT = Cfg::WPT, NSUPER = Cfg::NSUPER, ACC_CAP = Cfg::ACC_CAP, K_CAP = Cfg::K_CAP;
    __shared__ unsigned long long bm[WORDS];                 // the window's structure: one bit per column
    __shared__ uint16_t sub[NUMERIC ? WORDS : 1];            // outputs before a word inside its superblock
    __shared__ uint32_t super[NUMERIC ? NSUPER + 1 : 1];     // outputs before each 2048-column superblock
    __shared__ double acc[NUMERIC ? ACC_CAP : 1];            // accumulators of the current pass
    __shared__ uint64_t kS[K_CAP];
    __shared__ uint32_t kP[K_CAP + 1];
    __shared__ double kA[NUMERIC ? K_CAP : 1];
    __shared__ uint64_t wt[16];
    __shared__ uint32_t token;
    const uint32_t tid = threadIdx.x;
    const uint64_t t = large_list[task_of_block(blockIdx.x, gridDim.x, xcd_chunk)];
    const uint64_t r = task_row[t];
    const uint64_t jt = t - first_task[r], nt = ntasks[r];
    constexpr uint64_t W = 1ull << WL;
    uint64_t nwin = (b_cols + W - 1) >> WL;
    if (nwin == 0) nwin = 1;
    const uint64_t wpt = (nwin + nt - 1) / nt;               // windows per task of this row (row_work_kernel)
    const uint64_t w_begin = jt * wpt;
    uint64_t w_end = w_begin + wpt;
    if (w_end > nwin) w_end = nwin;
    const uint64_t as = (uint64_t)A.indptr[r], ae = (uint64_t)A.indptr[r + 1];
    const bool one_group = ae - as <= (uint64_t)K_CAP;
    const bool values = NUMERIC && c_data != nullptr;
    const bool lds_atomic = (flags & 1u) != 0;
    const bool retain_ok = values && (flags & 2u) != 0;
    // a row whose k's fit one staged group: thread j keeps k_j, the bounds of B's row k_j and a_ik for the whole task
    const bool mine_k = one_group && tid < (uint32_t)(ae - as) && w_begin < w_end;
    uint64_t rk = 0, rs = 0, re = 0, cur = 0, nxt_e = 0;
    double rav = 0.0;
    if (mine_k) {
        rk = (uint64_t)A.indices[as + tid];
        rs = (uint64_t)B.indptr[rk];
        re = (uint64_t)B.indptr[rk + 1];
        if (values) rav = A.data[as + tid];
        cur = w_begin == 0 ? rs : first_ge(B, rk, rs, rs, re, w_begin << WL);
        nxt_e = w_begin + 1 >= nwin ? re : first_ge(B, rk, rs, cur, re, (w_begin + 1) << WL);
    }
    if (tid == 0) token = 0;
    for (int i = tid; i < WORDS; i += LG_BLOCK) bm[i] = 0;
    lds_barrier();
    uint64_t out = 0;
    if constexpr (NUMERIC) out = off[t];
    uint32_t fresh = 0, tok_base = 0;
    (void)tok_base;
    for (uint64_t w = w_begin; w < w_end; ++w) {
        const uint64_t wlo = w << WL, whi = wlo + W;         // whi is not clamped to b_cols: see first_ge
        const uint64_t wcols = (whi < b_cols ? whi : b_cols) - wlo;
        const int words = (int)((wcols + SUPER_WORDS * 64 - 1) / (SUPER_WORDS * 64)) * SUPER_WORDS;   // whole superblocks
        // my k's sub-range in this window; the bound of the next window is requested now and used one window later
        const uint64_t ws = cur, we = nxt_e;
        if (mine_k) {
            cur = we;
            if (w + 1 < w_end) nxt_e = w + 2 >= nwin ? re : first_ge(B, rk, rs, we, re, (w + 2) << WL);
        }
        // ---- bit pass -------------------------------------------------------------------------------
        uint32_t k_total = 0;
        bool any = false, retain = false;
        Batch kept;
        for (uint64_t kc = as; kc < ae; kc += K_CAP) {
            const uint32_t n = (ae - kc < (uint64_t)K_CAP) ? (uint32_t)(ae - kc) : (uint32_t)K_CAP;
            const uint32_t gtot =
                one_group ? stage_compact<K_CAP>(mine_k ? (uint32_t)(we - ws) : 0u, ws, rav, kS, kP, values ? kA : (double *)nullptr, wt)
                          : stage_k_group<K_CAP>(A, B, kc, n, wlo, whi, nwin == 1, kS, kP, values ? kA : (double *)nullptr, wt);
            k_total = gtot;
            any |= gtot != 0;
            if constexpr (NUMERIC) {
                // A window of few entries (one batch per wave at most) is loaded ONCE: column, value and owner stay in
                // registers from the bit pass to the adds (its outputs fit one pass: <= 64 LG_U LG_WAVES <= ACC_CAP).
                retain = retain_ok && one_group && gtot <= 64u * LG_U * LG_WAVES;     // block-uniform
                if (retain) {
                    const uint32_t U = batch_u(gtot), wave = tid / WAVE;
#pragma unroll
                    for (int u = 0; u < LG_U; ++u) kept.val[u] = false;
                    if (wave * 64 * U < gtot) {
                        batch_load<K_CAP, true>(kept, B.indices, B.data, wlo, gtot, U, wave, kS, kP, kA);
                        batch_bits(kept, (uint32_t *)bm);
                    }
                    continue;
                }
            }
            walk_bits<K_CAP>(B.indices, wlo, gtot, kS, kP, (uint32_t *)bm);
        }
        lds_barrier();
        if (!any) continue;                                  // block-uniform; the bitmap is still clear
        if constexpr (!NUMERIC) {
            for (int i = tid; i < words; i += LG_BLOCK) {
                fresh += (uint32_t)__popcll(bm[i]);
                bm[i] = 0;
            }
            lds_barrier();
        } else {
            // ---- popcount prefix ------------------------------------------------------------------------
            // Words are dealt to the threads INTERLEAVED (thread tid takes words tid, tid + 512, ...): consecutive lanes
            // read consecutive LDS words, and the dense low columns of a power-law window are spread over all threads.
            // A superblock is 32 words = half a wave: its inner prefix is a 32-lane shuffle scan.
            static_assert(SUPER_WORDS == 32, "superblock = half a wave");
#pragma unroll 1
            for (int i = 0; i < WPT; ++i) {
                const int word = i * LG_BLOCK + (int)tid;
                if (i * LG_BLOCK >= words) break;                       // block-uniform
                const uint32_t pc = word < words ? (uint32_t)__popcll(bm[word]) : 0u;
                uint32_t inc = pc;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const uint32_t v = __shfl_up(inc, o, 32);
                    if ((tid & 31u) >= (uint32_t)o) inc += v;
                }
                if (word < words) {
                    sub[word] = (uint16_t)(inc - pc);
                    if ((tid & 31u) == 31u) super[word / SUPER_WORDS] = inc;   // superblock total, turned into a prefix below
                }
            }
            lds_barrier();
            const int nsb = words / SUPER_WORDS;                   // <= NSUPER <= 512 = one per thread
            uint32_t wtot;
            const uint32_t spre = block_excl_scan_u32_lds((int)tid < nsb ? super[tid] : 0u, (uint32_t *)wt, &wtot);
            if ((int)tid < nsb) super[tid] = spre;
            if (tid == 0) super[nsb] = wtot;
            lds_barrier();
            // indices come out sorted: walk the set bits of each word in order
            if (c_indices) {                                        // (null: C already has its structure)
#pragma unroll 1
                for (int i = 0; i < WPT; ++i) {
                    const int word = i * LG_BLOCK + (int)tid;
                    if (word >= words) break;
                    unsigned long long m = bm[word];
                    uint32_t run = super[word / SUPER_WORDS] + sub[word];
                    while (m) {
                        const int bit = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        c_indices[out + run] = (IDX)(wlo + (uint64_t)word * 64 + (uint64_t)bit);
                        ++run;
                    }
                }
            }
            // ---- values ---------------------------------------------------------------------------------
            // PASSES: the window's superblocks are cut greedily into ranges [pb, pe) of at most ACC_CAP outputs
            uint32_t pb = 0;
            while (values && pb < (uint32_t)nsb) {
                uint32_t pe = pb + 1;
                while (pe < (uint32_t)nsb && super[pe + 1] - super[pb] <= (uint32_t)ACC_CAP) ++pe;
                const uint32_t base_rank = super[pb];
                const uint32_t pass_out = super[pe] - base_rank;
                if (pass_out) {                                  // block-uniform
                    for (uint32_t i = tid; i < pass_out; i += LG_BLOCK) acc[i] = 0.0;   // tmp starts at N::zero()
                    const bool single = pb == 0 && pe == (uint32_t)nsb;
                    const uint64_t plo = wlo + (uint64_t)pb * (SUPER_WORDS * 64);
                    const uint64_t phi = wlo + (uint64_t)pe * (SUPER_WORDS * 64);
                    if (retain) {                                // (then single: see the bit pass)
                        lds_barrier();                           // the accumulators are clear before anybody adds
                        const uint32_t U = batch_u(k_total), wave = tid / WAVE;
                        const uint32_t nbatch = (k_total + 64 * U - 1) / (64 * U);
                        if (wave < nbatch) batch_add(kept, U, bm, sub, super, base_rank, acc, &token, tok_base + wave, lds_atomic);
                        tok_base += nbatch;
                    } else
                    for (uint64_t kc = as; kc < ae; kc += K_CAP) {
                        const uint32_t n = (ae - kc < (uint64_t)K_CAP) ? (uint32_t)(ae - kc) : (uint32_t)K_CAP;
                        uint32_t gtot;
                        if (single && one_group) {
                            gtot = k_total;                      // kS / kP / kA as the bit pass left them
                            lds_barrier();                       // the accumulators are clear before anybody adds
                        } else if (one_group) {
                            uint64_t s = ws, e = ws;
                            if (mine_k && we > ws) {
                                s = pb == 0 ? ws : first_ge(B, rk, rs, ws, we, plo);
                                e = pe == (uint32_t)nsb ? we : first_ge(B, rk, rs, s, we, phi);
                            }
                            gtot = stage_compact<K_CAP>((uint32_t)(e - s), s, rav, kS, kP, kA, wt);
                        } else {
                            gtot = stage_k_group<K_CAP>(A, B, kc, n, single ? wlo : plo, single ? whi : phi, single && nwin == 1,
                                                        kS, kP, kA, wt);
                        }
                        tok_base += walk_values<K_CAP>(B.indices, B.data, wlo, gtot, kS, kP, kA, bm, sub, super, base_rank, acc,
                                                       &token, tok_base, lds_atomic);
                    }
                    lds_barrier();                               // every add has been performed
                    for (uint32_t i = tid; i < pass_out; i += LG_BLOCK) c_data[out + base_rank + i] = acc[i];
                    lds_barrier();                               // the next pass clears the accumulators
                }
                pb = pe;
            }
            out += wtot;
            for (int i = tid; i < words; i += LG_BLOCK) bm[i] = 0;
            lds_barrier();
        }
    }
    if constexpr (!NUMERIC) {
        const uint64_t wsum = wave_sum_u64(fresh);
        if ((tid & (WAVE - 1)) == 0) wt[tid / WAVE] = wsum;
        lds_barrier();
        if (tid == 0) {
            uint64_t tot = 0;
            for (int i = 0; i < LG_WAVES; ++i) tot += wt[i];
            count[t] = tot;
        }
    }
}

template <typename PTR>
__global__ void write_indptr_kernel(const uint64_t *__restrict__ first_task, const uint64_t *__restrict__ off,
                                    uint64_t rows, PTR *__restrict__ indptr) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    indptr[r] = (PTR)off[first_task[r]];
}

// numeric into an existing matrix: its indptr must be the product's
template <typename PTR>
__global__ void compare_indptr_kernel(const uint64_t *__restrict__ first_task, const uint64_t *__restrict__ off,
                                      uint64_t rows, const PTR *__restrict__ indptr, unsigned int *__restrict__ mismatch) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    if ((uint64_t)indptr[r] != off[first_task[r]]) atomicOr(mismatch, 1u);
}

// temporaries come from the library's block pool (abi.hip): a product per iteration no longer pays ~15 hipMalloc / hipFree
struct DevBuf {
    void *p = nullptr;
    uint64_t cap = 0;
    int dev = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) pool_free(p, cap, dev);
        p = nullptr;
    }
    hipError_t alloc(uint64_t bytes) {
        release();
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        return pool_alloc(&p, bytes ? bytes : 8, &cap, dev);
    }
    template <typename T>
    T *as() { return (T *)p; }
};

}  // namespace

int32_t radix_sort_pairs(uint64_t *keys, uint64_t *vals, uint64_t n, const std::vector<std::pair<int, int>> &fields, hipStream_t stream);   // sort.hip

}  // namespace sprs_hip

// The symbolic phase of a product, kept: per-task output counts and offsets, the task lists, the column-bucket table of
// B.  What smmp::symbolic hands to smmp::numeric in the reference is C's indptr and indices (smmp.rs:81-131, 151-189);
// here the plan additionally remembers how the work was cut, so that numeric launches the value kernels only.
struct sprs_hip_spgemm_plan {
    int32_t idx_bytes = 8, iptr_bytes = 8;
    uint64_t rows = 0, inner = 0, b_cols = 0, nnz_a = 0, nnz_b = 0;
    const void *a_indptr = nullptr, *a_indices = nullptr, *b_indptr = nullptr, *b_indices = nullptr;   // whose structure it describes
    uint64_t ntask_total = 0, n_small = 0, n_large = 0, n_tiny = 0, c_nnz = 0, nb = 0;
    int64_t winlog = 17;
    uint32_t xcd_chunk = 0;        // how the launch deals the task list to the XCDs (task_of_block)
    sprs_hip::DevBuf bucket, ub, ntasks, first_task, task_row, tiny_list, small_list, large_list, count, off;
};

namespace sprs_hip {

namespace {

template <typename IDX, typename PTR>
CsrView<IDX, PTR> view_of(const sprs_hip_csmat *m) {
    return CsrView<IDX, PTR>{(const PTR *)m->indptr, (const IDX *)m->indices, m->data, nullptr, 0};
}

// ---- symbolic phase: counts, offsets, task lists ---------------------------------------------------------
template <typename IDX, typename PTR>
int32_t plan_build(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_spgemm_plan *pl) {
    hipStream_t stream = nullptr;
    const uint64_t rows = a->rows, b_cols = b->cols;
    pl->idx_bytes = (int32_t)sizeof(IDX);
    pl->iptr_bytes = (int32_t)sizeof(PTR);
    pl->rows = rows;
    pl->inner = a->cols;
    pl->b_cols = b_cols;
    pl->nnz_a = a->nnz;
    pl->nnz_b = b->nnz;
    pl->a_indptr = a->indptr;
    pl->a_indices = a->indices;
    pl->b_indptr = b->indptr;
    pl->b_indices = b->indices;
    pl->winlog = options().spgemm_winlog;
    pl->xcd_chunk = options().spgemm_xcd_chunk < 0 ? 0xFFFFFFFFu : (uint32_t)options().spgemm_xcd_chunk;
    CsrView<IDX, PTR> A = view_of<IDX, PTR>(a), B = view_of<IDX, PTR>(b);
    // column-bucket table of B (4 bytes per 2048 columns per row; config 5: 1.96 GB): bounded — up to 4 GiB outright
    // (1.4 % of the device), beyond that only within a small multiple of B's own size and never above 8 GiB (a
    // 10M x 10M operand would ask for 195 GB) — and skipped when pointless (every row of B has at most one entry);
    // on allocation failure: binary searches instead
    {
        const uint64_t nb = (b_cols >> BUCKET_LOG2) + 2;
        const uint64_t bytes = b->rows * nb * sizeof(uint32_t);
        const uint64_t b_bytes = b->nnz * (8 + sizeof(IDX)) + (b->rows + 1) * sizeof(PTR);
        if (options().spgemm_bucket && b->rows && b->nnz > b->rows && bytes <= (8ull << 30) &&
            (bytes <= (4ull << 30) || bytes <= 4 * b_bytes + (64ull << 20))) {
            if (pl->bucket.alloc(bytes) == hipSuccess) {
                uint64_t blocks = (b->rows + 3) / 4;
                if (blocks > 256 * 64) blocks = 256 * 64;
                hipLaunchKernelGGL((build_bucket_kernel<IDX, PTR>), dim3((unsigned)blocks), dim3(256), 0, stream, B.indptr,
                                   B.indices, b->rows, nb, pl->bucket.as<uint32_t>());
                SPRS_TRY_HIP(hipGetLastError());
                pl->nb = nb;
            } else {
                (void)hipGetLastError();
                clear_error();
            }
        }
    }
    B.bucket = pl->nb ? pl->bucket.as<uint32_t>() : nullptr;
    B.nb = pl->nb;

    DevBuf is_tiny, is_small, n_large_r, pos_tiny, pos_small, pos_large, large_key;
    SPRS_TRY_HIP(pl->ub.alloc(rows * 8));
    SPRS_TRY_HIP(pl->ntasks.alloc(rows * 8));
    SPRS_TRY_HIP(pl->first_task.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(is_tiny.alloc(rows * 8));
    SPRS_TRY_HIP(is_small.alloc(rows * 8));
    SPRS_TRY_HIP(n_large_r.alloc(rows * 8));
    SPRS_TRY_HIP(pos_tiny.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(pos_small.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(pos_large.alloc((rows + 1) * 8));
    const dim3 rgrid((unsigned)((rows + 255) / 256)), rblock(256);
    if (rows) {
        uint64_t blocks = (rows + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL((row_work_kernel<IDX, PTR>), dim3((unsigned)blocks), dim3(256), 0, stream, A, B, rows, b_cols,
                           (uint64_t)options().spgemm_heavy, (uint32_t)options().spgemm_winlog, pl->ub.as<uint64_t>(),
                           pl->ntasks.as<uint64_t>());
        SPRS_TRY_HIP(hipGetLastError());
        hipLaunchKernelGGL(task_class_kernel, rgrid, rblock, 0, stream, pl->ub.as<uint64_t>(), pl->ntasks.as<uint64_t>(), rows,
                           is_tiny.as<uint64_t>(), is_small.as<uint64_t>(), n_large_r.as<uint64_t>());
        SPRS_TRY_HIP(hipGetLastError());
    }
    SPRS_TRY(exclusive_scan_u64(pl->ntasks.as<uint64_t>(), pl->first_task.as<uint64_t>(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(is_tiny.as<uint64_t>(), pos_tiny.as<uint64_t>(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(is_small.as<uint64_t>(), pos_small.as<uint64_t>(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(n_large_r.as<uint64_t>(), pos_large.as<uint64_t>(), rows, stream));
    SPRS_TRY_HIP(hipMemcpy(&pl->ntask_total, pl->first_task.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&pl->n_tiny, pos_tiny.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&pl->n_small, pos_small.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&pl->n_large, pos_large.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost));
    const uint64_t ntask_total = pl->ntask_total, n_small = pl->n_small, n_large = pl->n_large, n_tiny = pl->n_tiny;
    if (n_large > 0x7fffffffull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "too many SpGEMM tasks for one launch");

    SPRS_TRY_HIP(pl->task_row.alloc(ntask_total * 8));
    SPRS_TRY_HIP(pl->tiny_list.alloc(n_tiny * 8));
    SPRS_TRY_HIP(pl->small_list.alloc(n_small * 8));
    SPRS_TRY_HIP(pl->large_list.alloc(n_large * 8));
    SPRS_TRY_HIP(large_key.alloc(n_large * 8));
    SPRS_TRY_HIP(pl->count.alloc(ntask_total * 8));
    SPRS_TRY_HIP(pl->off.alloc((ntask_total + 1) * 8));
    if (ntask_total) {
        hipLaunchKernelGGL(make_tasks_kernel, rgrid, rblock, 0, stream, pl->ntasks.as<uint64_t>(), pl->first_task.as<uint64_t>(),
                           pl->ub.as<uint64_t>(), rows, pos_tiny.as<uint64_t>(), pos_small.as<uint64_t>(), pos_large.as<uint64_t>(),
                           is_tiny.as<uint64_t>(), is_small.as<uint64_t>(), pl->task_row.as<uint64_t>(), pl->tiny_list.as<uint64_t>(),
                           pl->small_list.as<uint64_t>(), pl->large_list.as<uint64_t>(), large_key.as<uint64_t>());
        SPRS_TRY_HIP(hipGetLastError());
    }
    // costliest tasks first (stable sort by cost class); option spgemm_task_order = 2 keeps the row order (A/B)
    if (n_large > 1 && options().spgemm_task_order != 2)
        SPRS_TRY(radix_sort_pairs(large_key.as<uint64_t>(), pl->large_list.as<uint64_t>(), n_large, {{0, 6}}, stream));

    auto small_grid = [&](uint64_t n_tasks) {
        uint64_t g = (n_tasks + SM_WAVES - 1) / SM_WAVES;
        if (g > 256 * 32) g = 256 * 32;
        return dim3((unsigned)g);
    };
    if (n_tiny) {
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, false, TINY_TAB>), small_grid(n_tiny), dim3(SM_BLOCK), 0, stream,
                           A, B, pl->tiny_list.as<uint64_t>(), n_tiny, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (n_small) {
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, false, SMALL_TAB>), small_grid(n_small), dim3(SM_BLOCK), 0, stream,
                           A, B, pl->small_list.as<uint64_t>(), n_small, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr);
        SPRS_TRY_HIP(hipGetLastError());
    }
    // ONE launch for all large tasks, in the LDS layout of the window width (option spgemm_winlog)
    if (n_large) {
        const dim3 g((unsigned)n_large), blk(LG_BLOCK);
#define SPRS_LG_SYM(WL)                                                                                              \
    hipLaunchKernelGGL((large_rows_kernel<WL, IDX, PTR, false, 1>), g, blk, 0, stream, A, B, b_cols,                    \
                       pl->large_list.as<uint64_t>(), pl->task_row.as<uint64_t>(), pl->first_task.as<uint64_t>(),    \
                       pl->ntasks.as<uint64_t>(), pl->count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, \
                       (double *)nullptr, pl->xcd_chunk, 0u)
        switch (pl->winlog) {
            case 16: SPRS_LG_SYM(16); break;
            case 18: SPRS_LG_SYM(18); break;
            case 19: SPRS_LG_SYM(19); break;
            default: SPRS_LG_SYM(17); break;
        }
#undef SPRS_LG_SYM
        SPRS_TRY_HIP(hipGetLastError());
    }
    // ---- prefix sum of the counts -> offsets, C.indptr (smmp.rs:320-331) ----
    SPRS_TRY(exclusive_scan_u64(pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), ntask_total, stream));
    SPRS_TRY_HIP(hipMemcpy(&pl->c_nnz, pl->off.as<uint64_t>() + ntask_total, 8, hipMemcpyDeviceToHost));
    if (sizeof(PTR) == 4 && pl->c_nnz > 0xFFFFFFFFull)
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold the nnz of the product (%llu)",
                  (unsigned long long)pl->c_nnz);   // Iptr::from_usize, smmp.rs:121
    return SPRS_HIP_OK;
}

// ---- numeric phase: indices (optional) and values into a matrix of the product's structure -------------------------
template <typename IDX, typename PTR>
int32_t plan_run(sprs_hip_spgemm_plan *pl, const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c, bool values,
                 bool indices) {
    hipStream_t stream = nullptr;
    CsrView<IDX, PTR> A = view_of<IDX, PTR>(a), B = view_of<IDX, PTR>(b);
    B.bucket = pl->nb ? pl->bucket.as<uint32_t>() : nullptr;
    B.nb = pl->nb;
    const uint64_t n_tiny = pl->n_tiny, n_small = pl->n_small, n_large = pl->n_large;
    double *c_values = values ? c->data : nullptr;
    IDX *c_indices = indices ? (IDX *)c->indices : nullptr;
    auto small_grid = [&](uint64_t n_tasks) {
        uint64_t g = (n_tasks + SM_WAVES - 1) / SM_WAVES;
        if (g > 256 * 32) g = 256 * 32;
        return dim3((unsigned)g);
    };
    if (n_tiny)
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, true, TINY_TAB>), small_grid(n_tiny), dim3(SM_BLOCK), 0, stream, A,
                           B, pl->tiny_list.as<uint64_t>(), n_tiny, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), c_indices, c_values);
    if (n_small)
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, true, SMALL_TAB>), small_grid(n_small), dim3(SM_BLOCK), 0, stream,
                           A, B, pl->small_list.as<uint64_t>(), n_small, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), c_indices, c_values);
    if (n_large) {
        const dim3 g((unsigned)n_large), blk(LG_BLOCK);
        const uint32_t flags = (options().spgemm_lds_atomic ? 1u : 0u) | (options().spgemm_retain ? 2u : 0u);
#define SPRS_LG_NUM(WL, OCC)                                                                                         \
    hipLaunchKernelGGL((large_rows_kernel<WL, IDX, PTR, true, OCC>), g, blk, 0, stream, A, B, pl->b_cols,            \
                       pl->large_list.as<uint64_t>(), pl->task_row.as<uint64_t>(), pl->first_task.as<uint64_t>(),    \
                       pl->ntasks.as<uint64_t>(), pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), c_indices,       \
                       c_values, pl->xcd_chunk, flags)
        const bool occ3 = options().spgemm_occupancy != 2;
        switch (pl->winlog) {
            case 16: if (occ3) SPRS_LG_NUM(16, 6); else SPRS_LG_NUM(16, 4); break;
            case 18: SPRS_LG_NUM(18, 1); break;
            case 19: SPRS_LG_NUM(19, 1); break;
            default: if (occ3) SPRS_LG_NUM(17, 6); else SPRS_LG_NUM(17, 4); break;
        }
#undef SPRS_LG_NUM
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail_hip(e, "spgemm numeric");
    return SPRS_HIP_OK;
}

template <typename PTR>
int32_t plan_indptr(sprs_hip_spgemm_plan *pl, sprs_hip_csmat *c, bool compare) {
    hipStream_t stream = nullptr;
    const dim3 g((unsigned)((pl->rows + 256) / 256)), b(256);
    if (!compare) {
        hipLaunchKernelGGL((write_indptr_kernel<PTR>), g, b, 0, stream, pl->first_task.as<uint64_t>(), pl->off.as<uint64_t>(),
                           pl->rows, (PTR *)c->indptr);
        SPRS_TRY_HIP(hipGetLastError());
        return SPRS_HIP_OK;
    }
    DevBuf flag;
    SPRS_TRY_HIP(flag.alloc(4));
    SPRS_TRY_HIP(hipMemsetAsync(flag.p, 0, 4, stream));
    hipLaunchKernelGGL((compare_indptr_kernel<PTR>), g, b, 0, stream, pl->first_task.as<uint64_t>(), pl->off.as<uint64_t>(),
                       pl->rows, (const PTR *)c->indptr, flag.as<unsigned int>());
    unsigned int bad = 0;
    SPRS_TRY_HIP(hipMemcpy(&bad, flag.p, 4, hipMemcpyDeviceToHost));
    if (bad) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "numeric: the indptr of C is not the product's");
    return SPRS_HIP_OK;
}

#define SPRS_SPGEMM_DISPATCH(pl, CALL)                                                   \
    ((pl)->idx_bytes == 8 && (pl)->iptr_bytes == 8   ? CALL(uint64_t, uint64_t)          \
     : (pl)->idx_bytes == 4 && (pl)->iptr_bytes == 8 ? CALL(uint32_t, uint64_t)          \
     : (pl)->idx_bytes == 8                          ? CALL(uint64_t, uint32_t)          \
                                                     : CALL(uint32_t, uint32_t))

}  // namespace

int32_t spgemm_plan_create(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_spgemm_plan **out) {
    if (b->cols > 0xFFFFFFFEull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "SpGEMM: more than 2^32-2 columns is not supported");
    auto *pl = new sprs_hip_spgemm_plan();
    pl->idx_bytes = a->idx_bytes;
    pl->iptr_bytes = a->iptr_bytes;
#define SPRS_CALL(I, P) plan_build<I, P>(a, b, pl)
    const int32_t st = SPRS_SPGEMM_DISPATCH(pl, SPRS_CALL);
#undef SPRS_CALL
    if (st != SPRS_HIP_OK) {
        delete pl;
        return st;
    }
    *out = pl;
    return SPRS_HIP_OK;
}

void spgemm_plan_free(sprs_hip_spgemm_plan *pl) { delete pl; }

static int32_t plan_matches(const sprs_hip_spgemm_plan *pl, const sprs_hip_csmat *a, const sprs_hip_csmat *b) {
    if (pl->rows != a->rows || pl->inner != a->cols || pl->b_cols != b->cols || pl->nnz_a != a->nnz || pl->nnz_b != b->nnz ||
        pl->idx_bytes != a->idx_bytes || pl->iptr_bytes != a->iptr_bytes || pl->a_indptr != a->indptr || pl->a_indices != a->indices ||
        pl->b_indptr != b->indptr || pl->b_indices != b->indices)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "the plan was made for other operands (shape, index types or structure buffers differ)");
    return SPRS_HIP_OK;
}

// structure of the product as a new matrix (values zero): what smmp::symbolic returns
int32_t spgemm_plan_structure(sprs_hip_spgemm_plan *pl, const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **out,
                              bool with_values) {
    SPRS_TRY(plan_matches(pl, a, b));
    sprs_hip_csmat *c = nullptr;
    SPRS_TRY(alloc_csmat(&c, SPRS_HIP_CSR, pl->rows, pl->b_cols, pl->c_nnz, pl->iptr_bytes, pl->idx_bytes));
    int32_t st = pl->iptr_bytes == 8 ? plan_indptr<uint64_t>(pl, c, false) : plan_indptr<uint32_t>(pl, c, false);
    if (st == SPRS_HIP_OK && !with_values) {
        const hipError_t e = hipMemsetAsync(c->data, 0, (pl->c_nnz ? pl->c_nnz : 1) * sizeof(double), nullptr);
        if (e != hipSuccess) st = fail_hip(e, "spgemm structure");
    }
#define SPRS_CALL(I, P) plan_run<I, P>(pl, a, b, c, with_values, true)
    if (st == SPRS_HIP_OK) st = SPRS_SPGEMM_DISPATCH(pl, SPRS_CALL);
#undef SPRS_CALL
    if (st != SPRS_HIP_OK) {
        sprs_hip_csmat_free(c);
        return st;
    }
    *out = c;
    return SPRS_HIP_OK;
}

// values into a matrix that already has the product's structure (smmp::numeric, smmp.rs:151-189): shape, nnz and indptr
// are checked; the indices of c are neither read nor written
int32_t spgemm_plan_numeric(sprs_hip_spgemm_plan *pl, const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c) {
    SPRS_TRY(plan_matches(pl, a, b));
    if (c->nnz != pl->c_nnz)
        SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "numeric: C holds %llu entries, the product has %llu", (unsigned long long)c->nnz,
                  (unsigned long long)pl->c_nnz);
    SPRS_TRY(pl->iptr_bytes == 8 ? plan_indptr<uint64_t>(pl, c, true) : plan_indptr<uint32_t>(pl, c, true));
#define SPRS_CALL(I, P) plan_run<I, P>(pl, a, b, c, true, false)
    return SPRS_SPGEMM_DISPATCH(pl, SPRS_CALL);
#undef SPRS_CALL
}

uint64_t spgemm_plan_nnz(const sprs_hip_spgemm_plan *pl) { return pl->c_nnz; }

int32_t spgemm_f64(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c) {
    sprs_hip_spgemm_plan *pl = nullptr;
    SPRS_TRY(spgemm_plan_create(a, b, &pl));
    const int32_t st = spgemm_plan_structure(pl, a, b, c, true);
    spgemm_plan_free(pl);
    return st;
}

// smmp::symbolic (smmp.rs:81-131): structure only; the values of the result are zero
int32_t spgemm_symbolic(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c) {
    sprs_hip_spgemm_plan *pl = nullptr;
    SPRS_TRY(spgemm_plan_create(a, b, &pl));
    const int32_t st = spgemm_plan_structure(pl, a, b, c, false);
    spgemm_plan_free(pl);
    return st;
}

// smmp::numeric (smmp.rs:151-189) without a kept plan: the symbolic phase is redone to check c and to cut the work
int32_t spgemm_numeric(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c) {
    sprs_hip_spgemm_plan *pl = nullptr;
    SPRS_TRY(spgemm_plan_create(a, b, &pl));
    const int32_t st = spgemm_plan_numeric(pl, a, b, c);
    spgemm_plan_free(pl);
    return st;
}

}  // namespace sprs_hip
