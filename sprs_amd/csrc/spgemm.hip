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
// Work decomposition (row_work_kernel classes every row by its product count ub_i = sum_{k in A_i} nnz(B_k) and its k's):
//   * ub <= 512: ONE WAVE per row with an LDS hash table (keys + f64 accumulators): the products of the row are
//     walked 64 at a time in the reference's order, keys inserted in parallel, products added through order tags;
//     then a bitonic sort of the table in LDS.  Rows of <= 64 products use 128-slot tables (32 waves per CU).
//   * <= 64 k's and ub <= spgemm_mid (65 536): ONE WAVE per row walking column windows of 2^14 with an LDS bitmap
//     (mid_rows_kernel): no workgroup barrier anywhere, 20 independent waves per CU.
//   * the rest: a 512-thread WORKGROUP per row (per narrower window for a heavy row) walking windows of 2^17 columns
//     (large_rows_kernel).
//   In both window kernels setting bits is the symbolic pass, a popcount prefix over the bitmap turns a column into its
//   rank inside the (sorted!) output row — indices come out sorted for free — and the values are accumulated in LDS in
//   passes, in the reference's order: one ds_add_f64 per k-run of a wave instruction, the LDS executes a wave's
//   instructions in issue order, a token orders the waves of a workgroup (see "ORDER OF THE ADDITIONS" below).
//   * a column-bucket table of B (entries of every row before each 2048-column boundary, built per plan, 4 B per row
//     per 2048 columns) replaces the binary searches that locate a row inside a window / superblock range.
// Per-task counts are scanned (hand-written two-level prefix sum, scan.hip) into output offsets; C.indptr falls out of
// the same scan.  Integer / LDS / latency bound: no MFMA.
// History of what was measured (profiles/): per-row windows + LDS accumulators 1.96 s -> 0.50 s on
// config 5; bucket table, LDS staging, prefetch -> 0.454 s (waves owning column ranges, one k at a time); entry-parallel
// expansion with order tags -> 0.22 s (round 1); row tasks, LDS-ordered adds, wave-per-row kernel -> 0.112 s (round 2).
// Developer builds (make DEVTOOLS=1): option spgemm_prof prints per-class / per-phase timings of the numeric kernels; the
// release library compiles the timers and the printout out.
#include "common.hpp"
#include "lanes.hpp"
#include "scan.hpp"

#include <algorithm>
#include <type_traits>
#include <cstring>
#include <vector>

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint64_t SMALL_MAX = 512;       // products per row handled by the wave/hash path
constexpr int SMALL_TAB = 1024;           // hash slots per wave (load factor <= 0.5)
constexpr uint64_t TINY_MAX = 64;         // rows of at most this many products: same kernel with a 128-slot table,
constexpr int TINY_TAB = 128;             //   so that 32 waves share a CU instead of 12 (these rows are latency bound)
constexpr int SM_BLOCK = 128;             // 2 waves (28 KB of LDS per block at 1024 slots: 5 blocks per CU)
constexpr int SM_NBIN = 128;              // column bins of the rank pass that replaces a sort of the hash table
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
    // plan-owned copies of the right operand's entries for the window kernels (they are bound by the bytes they pull through
    // the fabric, not by instructions: profiles/r10b): the columns as 32-bit numbers (b_cols < 2^32) for the counting walks —
    // half the bytes of a usize index — and one 16-byte record {column, value} per entry for the value walks — one load
    // and ONE cache line per short run instead of two.  Null in views that do not come from a plan.
    const uint32_t *col32;
    const struct BRec *pack;
};

struct alignas(16) BRec {
    uint32_t col, pad;
    double val;
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
        if (e - s <= 64) {
            // a short row (nearly all of them): lane = BUCKET — t[bb] = its entries in lower buckets, counted over the row's
            // entries held one per lane, eight buckets per lane at a time — and the row's line of the table is written coalesced.
            // (Lane = entry, below, writes the run of buckets between two entries serially per lane: 60 uncoalesced 4-byte stores
            // per lane for a row of 8 entries under 490 buckets; config 5: 1.76 ms for the 1.96 GB table, profiles/r16z, r17b.)
            const uint32_t ne = (uint32_t)(e - s);
            const uint32_t myb = lane < ne ? (uint32_t)((uint64_t)indices[s + lane] >> BUCKET_LOG2) : 0xFFFFFFFFu;
            for (uint64_t bb0 = 0; bb0 < nb; bb0 += 512) {
                uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const uint32_t bbl = (uint32_t)bb0 + lane;                  // (nb < 2^32: b_cols < 2^43 — the table is bounded to 8 GiB anyway)
                for (uint32_t j = 0; j < ne; ++j) {                          // wave-uniform
                    const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)myb, (int)j);
#pragma unroll
                    for (int u = 0; u < 8; ++u) cnt[u] += bj < bbl + 64u * u ? 1u : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint64_t bb = bb0 + 64u * u + lane;
                    if (bb < nb) t[bb] = cnt[u];
                }
            }
            continue;
        }
        // entry q (first of its bucket b, previous entry in bucket pb < b) defines t[pb+1 .. b] = q
        for (uint64_t p = s + lane; p < e; p += 64) {
            const uint64_t b = (uint64_t)indices[p] >> BUCKET_LOG2;
            const int64_t pb = p > s ? (int64_t)((uint64_t)indices[p - 1] >> BUCKET_LOG2) : -1;
            for (int64_t bb = pb + 1; bb <= (int64_t)b; ++bb) t[bb] = (uint32_t)(p - s);
        }
        const int64_t lastb = (int64_t)((uint64_t)indices[e - 1] >> BUCKET_LOG2);
        for (uint64_t bb = (uint64_t)(lastb + 1) + lane; bb < nb; bb += 64) t[bb] = (uint32_t)(e - s);
    }
}

template <typename IDX>
__global__ __launch_bounds__(256) void pack_cols_kernel(const IDX *__restrict__ indices, uint64_t nnz, uint32_t *__restrict__ col32) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) col32[p] = (uint32_t)indices[p];
}

template <typename IDX>
__global__ __launch_bounds__(256) void pack_entries_kernel(const IDX *__restrict__ indices, const double *__restrict__ data, uint64_t nnz,
                                                           BRec *__restrict__ pack) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) {
        BRec r;
        r.col = (uint32_t)indices[p];
        r.pad = 0;
        r.val = data[p];
        pack[p] = r;
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
// classes of a row by its product count (and k's): 0 none, 1 tiny and 2 small (one wave, hash table), 3 mid (one wave, column
// windows), 4 large (a workgroup); 5 / 6 / 7 MICRO rows — at most 16 / 32 / 64 products AND k's: 4 / 2 / 1 rows per wave, no LDS
// (micro_rows_kernel) — every row of the reference's own benchmark matrices (uniform density, 4 entries per row) is one
constexpr uint8_t CLS_MICRO16 = 5, CLS_MICRO32 = 6, CLS_MICRO64 = 7;
// extent word of an entry of A (micro rows): start of B's row k (40 bits) | its length (24 bits, saturated)
constexpr uint32_t EXT_SHIFT = 40;
constexpr uint64_t MICRO_KEY32_COLS = (1ull << 26) - 2;     // columns of B up to which (column << 6 | position) stays below 2^32 - 1
constexpr uint64_t EXT_START = (1ull << EXT_SHIFT) - 1ull, EXT_LEN_MAX = (1ull << (64 - EXT_SHIFT)) - 1ull;

// 16 lanes per row, four rows per wave (round 6: one wave per row left 60 lanes idle on the 4-entry rows of the reference's
// benchmark matrices and made this pass 0.5 ms of their 4.6 ms product; long rows just take more strides)
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void row_work_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B, uint64_t rows,
                                                       uint64_t b_cols, uint64_t heavy_products, uint32_t wl, uint32_t min_wl, uint64_t mid_max,
                                                       uint64_t *__restrict__ ub, uint64_t *__restrict__ ntasks,
                                                       uint8_t *__restrict__ cls, uint8_t *__restrict__ wlog, uint32_t micro,
                                                       uint64_t *__restrict__ ext) {
    const uint32_t lane = threadIdx.x & (WAVE - 1), sub = lane >> 4, sl = lane & 15u;
    const uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x / WAVE);
    for (uint64_t rb = w0 * 4; rb < rows; rb += nw * 4) {                // wave-uniform
        const uint64_t r = rb + sub;
        const bool ok = r < rows;
        const uint64_t s = ok ? (uint64_t)A.indptr[r] : 0, e = ok ? (uint64_t)A.indptr[r + 1] : 0;
        uint64_t acc = 0;
        // the extent of B's row behind every entry goes to `ext` for the micro rows' kernels (start | saturated length: a micro
        // row's k's have at most 64 entries each)
        constexpr uint64_t HUB = 2048;                                   // k's from which the whole wave walks a row
        const unsigned long long hubs = __ballot(sl == 0 && e - s >= HUB);
        if (hubs) {                                                      // wave-uniform; a handful of rows per matrix
            // a hub row (R-MAT 1M: 22 260 k's) walked by ONE 16-lane group was this pass's tail on config 5; all 64 lanes take it,
            // four strides of 64 in flight, and the row's owner gets the sum
            for (int g4 = 0; g4 < 4; ++g4) {
                if (!((hubs >> (16 * g4)) & 1ull)) continue;
                const uint64_t hs = __shfl(s, 16 * g4, WAVE), he = __shfl(e, 16 * g4, WAVE);
                uint64_t part = 0;
                for (uint64_t p = hs + lane; p < he; p += 256) {
                    uint64_t k[4], bs[4], be[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) k[u] = (uint64_t)A.indices[p + 64 * u < he ? p + 64 * u : hs];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bs[u] = (uint64_t)B.indptr[k[u]];
                        be[u] = (uint64_t)B.indptr[k[u] + 1];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (p + 64 * u < he) {
                            const uint64_t len = be[u] - bs[u];
                            part += len;
                            if (ext) ext[p + 64 * u] = bs[u] | ((len < EXT_LEN_MAX ? len : EXT_LEN_MAX) << EXT_SHIFT);
                        }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
                if (sub == (uint32_t)g4) acc = part;                     // (every lane of the owner group: the group sum below divides it out)
            }
        }
        const bool hub_row = ok && e - s >= HUB;
        if (hub_row) {
            acc = sl == 0 ? acc : 0;                                     // the group's reduction below adds the 16 lanes
        } else if (e - s <= 16) {                                        // one stride of the 16 lanes: nearly every row
            if (s + sl < e) {
                const uint64_t k = (uint64_t)A.indices[s + sl];
                const uint64_t bs = (uint64_t)B.indptr[k], len = (uint64_t)B.indptr[k + 1] - bs;
                acc = len;
                if (ext) ext[s + sl] = bs | ((len < EXT_LEN_MAX ? len : EXT_LEN_MAX) << EXT_SHIFT);
            }
        } else {
            // longer rows: four strides in flight, every load UNCONDITIONAL (a lane past the row's end reads the row's first entry
            // and drops it) — a gather under a lane's condition is a branch that waits for that one load, and the 22 260 k's of
            // the hub row of config 5 were 348 steps of five dependent round trips: the 1.14 ms of this pass
            for (uint64_t p = s + sl; p < e; p += 64) {
                uint64_t k[4], bs[4], be[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) k[u] = (uint64_t)A.indices[p + 16 * u < e ? p + 16 * u : s];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bs[u] = (uint64_t)B.indptr[k[u]];
                    be[u] = (uint64_t)B.indptr[k[u] + 1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p + 16 * u < e) {
                        const uint64_t len = be[u] - bs[u];
                        acc += len;
                        if (ext) ext[p + 16 * u] = bs[u] | ((len < EXT_LEN_MAX ? len : EXT_LEN_MAX) << EXT_SHIFT);
                    }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);   // the 16 lanes of the row
        if (ok && sl == 0) {
            ub[r] = acc;
            // A large row is ONE task that walks its column windows (2^wl columns each) one after the other.
            uint64_t nt = acc ? 1 : 0;
            uint8_t c = !acc ? 0 : acc <= TINY_MAX ? 1 : acc <= SMALL_MAX ? 2 : (e - s <= 64 && acc <= mid_max) ? 3 : 4;
            if (c == 1 && micro) {
                const uint64_t nk = e - s;
                if (acc <= 16 && nk <= 16) c = CLS_MICRO16;
                else if (acc <= 32 && nk <= 32) c = CLS_MICRO32;
                else if (nk <= 64) c = CLS_MICRO64;
            }
            cls[r] = c;
            uint32_t wl_r = wl;
            if (c == 4 && acc > heavy_products) {
                // a heavy row (hub): narrower windows, one task per window, ~heavy_products products each on average, so that
                // it is not one serial chain at the end of the launch (its window 0 still is the longest task)
                uint64_t width = b_cols / (acc / heavy_products);
                wl_r = width <= 1 ? 0 : 63 - __clzll((long long)width);       // floor(log2)
                if (wl_r < min_wl) wl_r = min_wl;
                if (wl_r > wl) wl_r = wl;
                nt = (b_cols + (1ull << wl_r) - 1) >> wl_r;
                if (nt == 0) nt = 1;
            }
            wlog[r] = (uint8_t)wl_r;
            ntasks[r] = nt;
        }
    }
}

struct alignas(16) MicroRec {       // a micro row in its class list: its task (= slot of its count / offset) and the row itself
    uint64_t t, r;
};

// Task lists, deterministic (first version: atomicAdd tickets, i.e. an arbitrary order that changed from call to call).
// Classes of a row: tiny (<= 64 products), small (<= 512), mid, large (one task per column window), three micro classes.
// A row's place in its class list and its first task are PREFIX COUNTS over the rows: eight of them (tasks, large tasks, six
// class flags).  Rounds 1 - 5 wrote one 8-byte flag array per count and scanned each with the general scan (five scans of
// three launches, five 8-byte read-backs: 0.44 of the 1.72 ms of the reference's 2.5 M-row benchmark product, profiles/r16p).
// Now the counts are taken straight from the class bytes: blocks of CLS_RB consecutive rows are summed (class_counts_kernel<false>),
// one workgroup scans the block sums (class_sums_kernel; ONE read-back of the eight totals sizes the lists), and the second pass
// over the same blocks (class_counts_kernel<true>) rebuilds each row's prefix — ballots for the flags, a wave scan for the
// task counts — and writes the lists.
constexpr int CLS_NV = 8;                           // tasks, large tasks, tiny, small, mid, micro16, micro32, micro64

struct ClassLists {
    uint64_t *task_row, *first_task, *large_slot, *tiny_list, *small_list, *mid_list, *large_list, *large_key, *mid_key;
    MicroRec *m16, *m32, *m64;
};

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint64_t o = __shfl_up(v, off, WAVE);
        if (lane >= (uint32_t)off) v += o;
    }
    return v;
}

// grid: one block of 256 threads per CLS_RB = rb consecutive rows (rb a multiple of 256).  LISTS = false: sums[v * nblocks + block]
// = the block's eight counts.  LISTS = true: sums holds the exclusive prefix over the blocks; every row gets its positions.
template <bool LISTS>
__global__ __launch_bounds__(256) void class_counts_kernel(const uint8_t *__restrict__ cls, const uint64_t *__restrict__ ntasks,
                                                           const uint64_t *__restrict__ ub, uint64_t rows, uint64_t rb,
                                                           uint64_t *__restrict__ sums, ClassLists out) {
    __shared__ uint64_t wtot[2][256 / WAVE][CLS_NV];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const uint64_t nblocks = gridDim.x, row0 = (uint64_t)blockIdx.x * rb;
    uint64_t base[CLS_NV];
#pragma unroll
    for (int v = 0; v < CLS_NV; ++v) base[v] = LISTS ? sums[(uint64_t)v * nblocks + blockIdx.x] : 0;
    const uint8_t want[6] = {1, 2, 3, CLS_MICRO16, CLS_MICRO32, CLS_MICRO64};
    if constexpr (!LISTS) {
        // sums only: every thread counts its own rows, one LDS addition per thread and count at the end
        __shared__ unsigned long long tot[CLS_NV];
        if (threadIdx.x < CLS_NV) tot[threadIdx.x] = 0;
        __syncthreads();
        for (uint64_t r = row0 + threadIdx.x; r < row0 + rb && r < rows; r += 256) {
            const uint8_t c = cls[r];
            const uint64_t nt = ntasks[r];
            base[0] += nt;
            base[1] += c == 4 ? nt : 0;
#pragma unroll
            for (int v = 0; v < 6; ++v) base[2 + v] += c == want[v] ? 1u : 0u;
        }
#pragma unroll
        for (int v = 0; v < CLS_NV; ++v)
            if (base[v]) atomicAdd(&tot[v], (unsigned long long)base[v]);
        __syncthreads();
        if (threadIdx.x < CLS_NV) sums[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = tot[threadIdx.x];
        return;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t it = 0;
    for (uint64_t r0 = row0; r0 < row0 + rb && r0 < rows; r0 += 256, ++it) {       // block-uniform
        const uint64_t r = r0 + threadIdx.x;
        const bool ok = r < rows;
        const uint8_t c = ok ? cls[r] : (uint8_t)0;
        const uint64_t nt = ok ? ntasks[r] : 0;
        uint64_t pre[CLS_NV], tot[CLS_NV];
        {
            const uint64_t inc = wave_incl_scan_u64(nt);
            pre[0] = inc - nt;
            tot[0] = __shfl(inc, WAVE - 1, WAVE);
            const uint64_t nl = c == 4 ? nt : 0, incl = wave_incl_scan_u64(nl);
            pre[1] = incl - nl;
            tot[1] = __shfl(incl, WAVE - 1, WAVE);
        }
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const unsigned long long m = __ballot(c == want[v]);
            pre[2 + v] = (uint64_t)__popcll(m & below);
            tot[2 + v] = (uint64_t)__popcll(m);
        }
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < CLS_NV; ++v) wtot[it & 1][wave][v] = tot[v];
        }
        __syncthreads();                                                           // (two buffers: one barrier per 256 rows)
#pragma unroll
        for (int v = 0; v < CLS_NV; ++v) {
            uint64_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < 256 / WAVE; ++w) {
                const uint64_t t = wtot[it & 1][w][v];
                if ((uint32_t)w < wave) before += t;
                all += t;
            }
            pre[v] += base[v] + before;
            base[v] += all;
        }
        if constexpr (LISTS) {
            if (ok) {
                const uint64_t f = pre[0];
                out.first_task[r] = f;
                out.large_slot[r] = pre[1];
                for (uint64_t j = 0; j < nt; ++j) out.task_row[f + j] = r;
                if (nt) {
                    if (c == 1) {
                        out.tiny_list[pre[2]] = f;
                    } else if (c == 2) {
                        out.small_list[pre[3]] = f;
                    } else if (c == 3) {
                        out.mid_list[pre[4]] = f;
                        out.mid_key[pre[4]] = (uint64_t)__clzll((long long)(ub[r] | 1));
                    } else if (c == CLS_MICRO16) {
                        out.m16[pre[5]] = MicroRec{f, r};
                    } else if (c == CLS_MICRO32) {
                        out.m32[pre[6]] = MicroRec{f, r};
                    } else if (c == CLS_MICRO64) {
                        out.m64[pre[7]] = MicroRec{f, r};
                    } else {
                        // lists in row order; for the large tasks also the sort key: the cost class (log2 of the products per task),
                        // costliest first, so that the long tasks start early and the short ones fill the tail of the launch (the
                        // sort is stable: inside a class the tasks stay in row order, and the list is the same run to run)
                        const uint64_t pos = pre[1];
                        const uint64_t cost = ub[r] / nt;
                        const uint64_t key = (uint64_t)__clzll((long long)(cost | 1));     // 0 .. 63, small = costly
                        for (uint64_t j = 0; j < nt; ++j) {
                            out.large_list[pos + j] = f + j;
                            out.large_key[pos + j] = key;
                        }
                    }
                }
            }
        }
    }
}

// one workgroup of CLS_NV waves: wave v turns sums[v * nblocks ..] into its exclusive prefix and leaves the total in totals[v];
// the total of the tasks also closes first_task (first_task[rows] = number of tasks)
__global__ __launch_bounds__(CLS_NV *WAVE) void class_sums_kernel(uint64_t *__restrict__ sums, uint64_t nblocks, uint64_t *__restrict__ totals,
                                                                  uint64_t *__restrict__ first_task, uint64_t rows) {
    const uint32_t lane = threadIdx.x & (WAVE - 1), v = threadIdx.x / WAVE;
    uint64_t *s = sums + (uint64_t)v * nblocks;
    uint64_t carry = 0;
    uint64_t xn = lane < nblocks ? s[lane] : 0;                                     // (the next chunk is requested while this one is scanned)
    for (uint64_t b0 = 0; b0 < nblocks; b0 += WAVE) {                               // wave-uniform
        const uint64_t b = b0 + lane;
        const uint64_t x = xn;
        xn = b + WAVE < nblocks ? s[b + WAVE] : 0;
        const uint64_t inc = wave_incl_scan_u64(x);
        if (b < nblocks) s[b] = carry + inc - x;
        carry += __shfl(inc, WAVE - 1, WAVE);
    }
    if (lane == 0) {
        totals[v] = carry;
        if (v == 0) first_task[rows] = carry;
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
                                                              IDX *__restrict__ c_indices, double *__restrict__ c_data,
                                                              uint32_t bin_shift, uint32_t flags) {
    __shared__ uint32_t keys_s[SM_WAVES][TAB];
    __shared__ double vals_s[NUMERIC ? SM_WAVES : 1][NUMERIC ? TAB : 1];
    __shared__ uint32_t tag_s[NUMERIC ? SM_WAVES : 1][WAVE];   // order tags of the entry-parallel path
    __shared__ uint32_t bin_s[NUMERIC ? SM_WAVES : 1][NUMERIC ? SM_NBIN : 1];      // rank pass: keys per column bin, then the bins' ends
    __shared__ uint32_t skey_s[NUMERIC ? SM_WAVES : 1][NUMERIC ? TAB / 2 : 1];     // rank pass: the keys grouped by bin
    const bool lane_order = (flags & 1u) != 0;                 // one ds_add_f64 per wave instruction: see add_lanes()
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
                const uint64_t pos = valid ? bs_o + (uint64_t)(t - (inc_o - len_o)) : 0ull;   // (a lane without an entry loads entry 0 and drops it:
                const uint32_t c = (uint32_t)B.indices[pos];                                   //  a load under a per-lane condition ends in its own wait)
                double pr = 0.0;
                if constexpr (NUMERIC) pr = av_o * B.data[pos];
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
                    if (lane_order) {                         // the lanes hold consecutive positions: the LDS applies them in lane order
                        if (valid) atomicAdd(&vals[h], pr);
                        wave_sync_lds();
                        pend = false;
                    }
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
            // The row comes out SORTED without sorting the table: the keys are counted per column bin (SM_NBIN bins of 2^bin_shift
            // columns), grouped by bin, and a key's rank is its bin's start plus the keys of its bin that are smaller — a short scan
            // of the bin.  (Rounds 1 to 3: a bitonic sort of the whole table, 45 stages of 8 dependent LDS round trips for a row of
            // 200 outputs — most of this kernel's time.)
            uint32_t *bin = bin_s[wave], *skey = skey_s[wave];
#pragma unroll
            for (int i = 0; i < SM_NBIN / WAVE; ++i) bin[i * WAVE + lane] = 0;
            wave_sync_lds();
            for (uint32_t i = lane; i < tsize; i += WAVE) {
                const uint32_t c = keys[i];
                if (c != EMPTY) atomicAdd(&bin[c >> bin_shift], 1u);
            }
            wave_sync_lds();
            {   // exclusive scan of the bin counts (two bins per lane) -> where each bin starts
                static_assert(SM_NBIN == 2 * WAVE, "two bins per lane");
                const uint32_t a0 = bin[2 * lane], a1 = bin[2 * lane + 1];
                uint32_t inc = a0 + a1;
#pragma unroll
                for (int off2 = 1; off2 < WAVE; off2 <<= 1) {
                    const uint32_t o2 = __shfl_up(inc, off2, WAVE);
                    if (lane >= (uint32_t)off2) inc += o2;
                }
                bin[2 * lane] = inc - a0 - a1;
                bin[2 * lane + 1] = inc - a1;
            }
            wave_sync_lds();
            for (uint32_t i = lane; i < tsize; i += WAVE) {
                const uint32_t c = keys[i];
                if (c != EMPTY) skey[atomicAdd(&bin[c >> bin_shift], 1u)] = c;     // afterwards bin[b] = END of bin b
            }
            wave_sync_lds();
            const uint64_t o = off[t];
            for (uint32_t i = lane; i < tsize; i += WAVE) {
                const uint32_t c = keys[i];
                if (c != EMPTY) {
                    const uint32_t b = c >> bin_shift;
                    const uint32_t e2 = bin[b];
                    uint32_t rank = b ? bin[b - 1] : 0u;
                    for (uint32_t j = rank; j < e2; ++j) rank += skey[j] < c ? 1u : 0u;
                    if (c_indices) c_indices[o + rank] = (IDX)c;      // null: C already has its structure (numeric on a kept plan)
                    if (c_data) c_data[o + rank] = vals[i];           // null: structure only (the twin of smmp::symbolic)
                }
            }
            wave_sync_lds();
        }
    }
}

// ---------------------------------------------------------------------------
// MICRO rows (round 6): G = 16 / 32 / 64 lanes per row, 64 / G rows per wave, no LDS, no hash table.
//
// The reference's own benchmark (sprs-benches/src/main.rs:148-163: uniform density, 4 entries per row, up to 2.5 M rows) is
// made of rows of ~16 products.  One wave per such row (small_rows_kernel) idles 48 lanes and walks a chain of nine dependent
// round trips per row: 2.1 + 1.5 ms of the 4.6 ms product (profiles/r15a).  Here a lane group owns a row:
//   * lane j of the group holds the extent of B's row k_j — start and length in one word, left per entry of A by
//     row_work_kernel, which had to fetch B.indptr[k], B.indptr[k + 1] anyway: a streamed load here instead of two more
//     dependent gathers per pass; a group-wide prefix of the lengths places the row's expansion — k ascending, columns
//     ascending inside a k: the reference's own order (smmp.rs:174-181) — one product per lane;
//   * the group SORTS its products by (column, position) — a bitonic network over the lanes, log2(G)(log2(G)+1)/2 exchanges
//     (the first version let every lane meet every other one: 2(G - 1) rotations, 854 against 527 us for the rows of 17 - 32
//     products, profiles/r16m).  Equal columns then sit side by side in ascending position: the first of a run is the output
//     entry, its rank is the number of runs before it (the row comes out sorted, smmp.rs:124), and it adds the rest of its run
//     one by one, from 0.0 + its own — the reference's chain (smmp.rs:166-181), bit for bit;
//   * symbolic: the number of runs.
// A row is a chain of dependent round trips (record -> bounds of the A row -> extents -> entries of B) and a wave has nothing
// else to do meanwhile, so the chain is a software pipeline over the wave's rows: while the entries of row i are in flight
// the extents of row i + 1, the bounds of row i + 2 and the record of row i + 3 are requested — one round trip per row.
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR, bool NUMERIC, int G, typename KEY>
__global__ __launch_bounds__(256) void micro_rows_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B, const uint64_t *__restrict__ ext,
                                                         const MicroRec *__restrict__ list, uint64_t n, uint64_t *__restrict__ count,
                                                         const uint64_t *__restrict__ off, IDX *__restrict__ c_indices,
                                                         double *__restrict__ c_data) {
    constexpr int R = WAVE / G, LOG2G = G == 16 ? 4 : G == 32 ? 5 : 6;
    constexpr KEY NO_KEY = ~(KEY)0;                                      // a lane without a product: the end of the order
    const uint32_t lane = threadIdx.x & (WAVE - 1), g = lane / G, gl = lane % G;
    const uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x / WAVE);
    const uint64_t stride = nw * R;
    auto load_rec = [&](uint64_t qq0) {                                  // (past the end: the last record again, dropped below)
        const uint64_t qq = qq0 + g;
        return list[qq < n ? qq : n - 1];
    };
    // every load is unconditional — a load under a lane's condition is a branch that ends in a wait for that one load (see the
    // window kernels): a lane without a k reads the row's first entry (a micro row has one) and drops it
    struct Bounds { uint64_t as, ae, o, t; };
    auto load_bounds = [&](const MicroRec &rc) {
        Bounds bd;
        bd.as = (uint64_t)A.indptr[rc.r];
        bd.ae = (uint64_t)A.indptr[rc.r + 1];
        bd.o = NUMERIC ? off[rc.t] : 0;
        bd.t = rc.t;
        return bd;
    };
    struct Ext { uint64_t e; double av; uint64_t o, t; uint32_t nk; };
    auto load_ext = [&](const Bounds &bd, uint64_t qq0) {
        Ext x;
        x.nk = qq0 + g < n ? (uint32_t)(bd.ae - bd.as) : 0u;            // <= G by the row's class
        const uint64_t at = bd.as + (gl < x.nk ? gl : 0u);
        x.e = ext[at];
        x.av = NUMERIC ? A.data[at] : 0.0;
        x.o = bd.o;
        x.t = bd.t;
        return x;
    };
    const uint64_t q00 = w0 * R;
    MicroRec rec2 = load_rec(q00 + 2 * stride);
    Bounds bd1 = load_bounds(load_rec(q00 + stride));
    Ext x0 = load_ext(load_bounds(load_rec(q00)), q00);
    for (uint64_t q0 = q00; q0 < n; q0 += stride) {                      // wave-uniform
        const MicroRec rec3 = load_rec(q0 + 3 * stride);
        const Bounds bd2 = load_bounds(rec2);
        const Ext x1 = load_ext(bd1, q0 + stride);
        const Ext xc = x0;                                               // (handed on: the body below works on row q0's copy)
        x0 = x1;
        bd1 = bd2;
        rec2 = rec3;
        const bool row_ok = q0 + g < n;
        const bool has = gl < xc.nk;
        const uint64_t bs = xc.e & EXT_START;
        const uint32_t len = has ? (uint32_t)(xc.e >> EXT_SHIFT) : 0u;
        const uint32_t inc = group_incl_scan_u32<G>(len);                // inclusive prefix of the lengths over the group (DPP, lanes.hpp)
        uint32_t total;                                                  // products of the row (<= G)
        if constexpr (G == WAVE) total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
        else total = __shfl(inc, G - 1, G);
        const bool valid = gl < total;                                   // lane = position gl of the expansion
        uint32_t own = 0;                                                // its k: the number of group lanes whose prefix is <= gl
#pragma unroll
        for (int step = G / 2; step > 0; step >>= 1) {
            const uint32_t v = __shfl(inc, (int)(own + step - 1), G);
            if (v <= gl) own += step;
        }
        own &= G - 1;
        // position gl of the expansion is entry (start of B's row) + gl - (products before that row): one word to fetch from the owner
        const uint64_t base_o = __shfl(bs - (uint64_t)(inc - len), (int)own, G);
        const uint64_t pos = valid ? base_o + gl : 0ull;                 // (a lane without a product loads entry 0 and drops it)
        uint32_t c;
        double pr = 0.0;
        if constexpr (NUMERIC) {
            const double av_o = __shfl(xc.av, (int)own, G);
            // ({column, value} records of B — one 64-byte gather per k instead of 16 + 32 — take 35 us off the three value kernels
            // of the 2.5 M-row benchmark product and cost 57 us to build: profiles/r16r)
            c = B.col32[pos];
            pr = av_o * B.data[pos];
        } else {
            c = B.col32[pos];
        }
        // the group sorts (column, position) keys; a key is one 32-bit word when B has fewer than 2^26 - 1 columns
        const KEY sk = group_sort<G, KEY>(valid ? ((KEY)c << LOG2G) | (KEY)gl : NO_KEY);
        const uint32_t cs = (uint32_t)(sk >> LOG2G), ps = (uint32_t)sk & (uint32_t)(G - 1);
        const uint32_t cprev = lane_below(cs);
        const bool head = sk != NO_KEY && (gl == 0 || cprev != cs);
        const unsigned long long hm = __ballot(head);
        const unsigned long long mine = G == WAVE ? hm : (hm >> (g * G)) & ((1ull << (G % WAVE)) - 1ull);
        if constexpr (!NUMERIC) {
            if (row_ok && gl == 0) count[xc.t] = (uint64_t)__popcll(mine);
        } else {
            const double v = __shfl(pr, (int)ps, G);
            const uint32_t rank = (uint32_t)__popcll(mine & ((1ull << gl) - 1ull));
            double acc = 0.0 + v;                                        // tmp starts at N::zero() (smmp.rs:166-170)
            if (__ballot(sk != NO_KEY && !head) != 0ull) {               // some column of the wave's rows has more than one product (wave-uniform)
                bool alive = head;
                for (int t = 1; t < G; ++t) {                            // (every lane takes part in every exchange; the exit is wave-uniform)
                    const uint32_t cn = __shfl_down(cs, (unsigned)t, G);
                    const double vn = __shfl_down(v, (unsigned)t, G);
                    alive = alive && gl + (uint32_t)t < (uint32_t)G && cn == cs;
                    if (__ballot(alive) == 0ull) break;
                    if (alive) acc += vn;
                }
            }
            if (head) {
                if (c_indices) c_indices[xc.o + rank] = (IDX)cs;         // null: C already has its structure (numeric on a kept plan)
                if (c_data) c_data[xc.o + rank] = acc;                   // null: structure only (the twin of smmp::symbolic)
            }
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
// and deterministic.  tests/test_spgemm_gpu.py compares bits with the CPU restatement of the reference; the CPU emulator (tests/emu) runs the
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

template <int K_CAP, bool VALUES>
__device__ __forceinline__ void batch_load(Batch &bt, const uint32_t *__restrict__ b_col32, const BRec *__restrict__ b_pack,
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
        pos[u] = 0;                            // a lane without an entry loads entry 0 and drops it: see the loads below
        bt.own[u] = 0;
        av[u] = 0.0;
        if (bt.val[u]) {
            wk.seek<K_CAP>(kS, kP, t);
            pos[u] = wk.base + t;
            bt.own[u] = wk.o;
            if constexpr (VALUES) av[u] = kA[wk.o];
        }
    }
    // All loads of the batch are independent and UNCONDITIONAL (a load under a per-lane condition is compiled as a branch
    // whose arm waits for that one load: four memory round trips per batch instead of one — the ISA of rounds 1 to 3).
    uint32_t col[LG_U];
    double bv[LG_U];
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        bv[u] = 0.0;
        if constexpr (VALUES) {                // one 16-byte record per entry
            const BRec rec = b_pack[pos[u]];
            col[u] = rec.col;
            bv[u] = rec.val;
        } else {
            col[u] = b_col32[pos[u]];
        }
    }
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        bt.cc[u] = bt.val[u] ? col[u] - (uint32_t)wlo : 0u;
        bt.pr[u] = av[u] * bv[u];
    }
}

__device__ __forceinline__ void batch_bits(const Batch &bt, uint32_t *bm32) {
#pragma unroll
    for (int u = 0; u < LG_U; ++u)
        if (bt.val[u]) atomicOr(&bm32[bt.cc[u] >> 5], 1u << (bt.cc[u] & 31));   // little endian: bit c & 63 of word c >> 6
}

template <int K_CAP>
__device__ __forceinline__ void walk_bits(const uint32_t *__restrict__ b_col32, uint64_t wlo, uint32_t gtot, const uint64_t *kS,
                                          const uint32_t *kP, uint32_t *bm32) {
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t U = batch_u(gtot);
    const uint32_t nbatch = (gtot + 64 * U - 1) / (64 * U);
    for (uint32_t b = wave; b < nbatch; b += LG_WAVES) {
        Batch bt;
        batch_load<K_CAP, false>(bt, b_col32, (const BRec *)nullptr, wlo, gtot, U, b, kS, kP, (const double *)nullptr);
        batch_bits(bt, bm32);
    }
}

// Volatile accesses to LDS words through explicitly LDS-qualified pointers: a volatile access through a GENERIC pointer is
// not narrowed by the compiler and becomes a flat load with system scope followed by s_waitcnt vmcnt(0) — every poll of
// the token then waited for all the global stores the wave had in flight (the index emission), first measured as 60 % of
// the kernel (profiles/r02u).
#ifdef SPRS_HIP_EMU
__device__ __forceinline__ uint32_t lds_load_u32(const uint32_t *p) { return *(const volatile uint32_t *)p; }
__device__ __forceinline__ void lds_store_u32(uint32_t *p, uint32_t v) { *(volatile uint32_t *)p = v; }
__device__ __forceinline__ double lds_load_f64(const double *p) { return *(const volatile double *)p; }
__device__ __forceinline__ void lds_store_f64(double *p, double v) { *(volatile double *)p = v; }
#else
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile double lds_vf64;
__device__ __forceinline__ uint32_t lds_load_u32(const uint32_t *p) { return *(const lds_vu32 *)p; }
__device__ __forceinline__ void lds_store_u32(uint32_t *p, uint32_t v) { *(lds_vu32 *)p = v; }
__device__ __forceinline__ double lds_load_f64(const double *p) { return *(const lds_vf64 *)p; }
__device__ __forceinline__ void lds_store_f64(double *p, double v) { *(lds_vf64 *)p = v; }
#endif

__device__ __forceinline__ void token_wait(uint32_t *token, uint32_t turn) {
    if (turn == 0xFFFFFFFFu) return;       // option spgemm_ordered = 0: no ordering of the waves' adds
    while (lds_load_u32(token) != turn) SPRS_POLL_PAUSE();
    asm volatile("" ::: "memory");
    wave_sync_lds();      // (no instruction; in the CPU emulator, where the lanes of a wave run one after the other, it keeps lane 0
                          // from passing the token on before its fellow lanes have seen their turn)
}

__device__ __forceinline__ void token_pass(uint32_t *token, uint32_t next) {
    if (next == 0u) return;                // (turn 0xFFFFFFFF + 1: the experiment above)
    SPRS_LDS_FENCE();                      // this wave's adds have been performed
    if ((threadIdx.x & (WAVE - 1)) == 0) lds_store_u32(token, next);
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
                lds_store_f64(acc + slot, lds_load_f64(acc + slot) + pr);   // A/B switch: read, add, write (same order)
            }
        }
        todo &= ~__ballot(mine);
    }
}

// ---------------------------------------------------------------------------
// mid rows: ONE WAVE per row, no workgroup barrier anywhere
//
// Rows of at most 64 k's and at most `mid_max` products (config 5: 257 000 of the 290 000 rows above the hash path, a
// third of the products).  As workgroup tasks they paid ~23 us per column window in barriers and dependent LDS round trips
// for a few hundred products (profiles/r02w: 57 of 99 s of workgroup time).  A wave needs no barrier: lane j keeps k_j, the
// bounds of B's row k_j and a_ik in registers; the row is walked in windows of 2^13 columns (128 bitmap words, 2 per lane;
// the bound of the next window comes from the bucket table one window ahead); a window is the bit pass, a popcount prefix
// by wave scans, the sorted indices, and the values in passes of 512 accumulators — windows of up to 256 entries keep them
// in registers in between.  The order of the additions is the LDS's own (one wave, instructions in order, one ds_add_f64
// per k-run: see large rows below).  24 independent waves per CU hide each other's memory round trips.
// ---------------------------------------------------------------------------
constexpr int MID_BLOCK = 128;
constexpr int MID_WAVES = MID_BLOCK / WAVE;
constexpr int MID_ACC = 512;                      // accumulators of one pass
constexpr int MID_K = 64;                         // k's per row: one per lane

// inclusive scan over the 64 lanes without the LDS: four row_shr steps inside the rows of 16 lanes, then row_bcast 15 and 31
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) { return group_incl_scan_u32<WAVE>(v); }   // lanes.hpp

// The products of one wave instruction into their accumulators.  The lanes hold 64 CONSECUTIVE positions of the expansion
// (k ascending with the lane), so lanes that meet in one accumulator must be applied in ascending lane order.
// lane_order = true: ONE ds_add_f64 — the LDS applies the lanes of an atomic instruction that hit one address in ascending
// lane order (measured on gfx950: scripts/probes/lds_add_order.hip, 2.2e7 sums of which 2.2e6 order-sensitive, every one equal
// to the ascending-lane sum; the library repeats a short form of that probe once per device, lds_lane_order_ok(), and falls
// back to one instruction per k-run when it does not hold).
__device__ __forceinline__ void add_lanes(bool val, uint32_t own, uint32_t slot, double pr, double *acc, bool lane_order) {
    if (lane_order) {
#ifdef SPRS_HIP_EMU
        wave_sync_lds();      // the CPU emulator runs a lane until its next collective: without this a lane would issue the adds of
                              // several wave instructions before its neighbours issue their first (the hardware is instruction-major)
#endif
        if (val) atomicAdd(&acc[slot], pr);                      // ds_add_f64, no return value
    } else {
        add_runs(val, own, slot, pr, acc, true);
    }
}

// what a position of the expansion needs from its k: where the k's sub-range starts (as the offset that turns a flat
// position into an entry of B) and a_ik — one 16-byte LDS read
struct alignas(16) KRec {
    uint64_t base;
    double a;
};

template <typename IDX, typename PTR, bool NUMERIC, int MID_WL, int MID_KEEP>
__global__ __launch_bounds__(MID_BLOCK, !NUMERIC ? 4 : MID_KEEP >= 8 ? 3 : 5) void mid_rows_kernel(CsrView<IDX, PTR> A, CsrView<IDX, PTR> B, uint64_t b_cols,
                                                             const uint64_t *__restrict__ mid_list, uint64_t n_mid,
                                                             const uint64_t *__restrict__ task_row,
                                                             uint64_t *__restrict__ count,        // symbolic: out
                                                             const uint64_t *__restrict__ off,    // numeric: in
                                                             IDX *__restrict__ c_indices, double *__restrict__ c_data,
                                                             unsigned long long *__restrict__ prof,
                                                             const uint64_t *__restrict__ ub_dbg,
                                                             unsigned int *__restrict__ next_row, uint32_t flags) {
    // The phase timers of option spgemm_prof (developer builds) live in LDS and are touched only when prof is set: kept in
    // registers (ten 64-bit values per lane) they cost the numeric kernel 92 bytes of scratch per lane.  The release library
    // compiles them out.
    constexpr bool TIMERS = DEVTOOLS;
    __shared__ unsigned long long ph_s[MID_BLOCK / WAVE][10];           // [wave][0..7 phases, 8 previous mark, 9 kernel start]
    unsigned long long *ph = ph_s[threadIdx.x / WAVE];
    if (TIMERS && prof && (threadIdx.x & (WAVE - 1)) == 0) {
        for (int i = 0; i < 8; ++i) ph[i] = 0;
        ph[8] = ph[9] = (unsigned long long)wall_clock64();
    }
    auto mark = [&](int phase) {                                        // lane 0's view of the phases (debug option spgemm_prof)
        if (TIMERS && prof && (threadIdx.x & (WAVE - 1)) == 0) {
            const unsigned long long now = (unsigned long long)wall_clock64();
            ph[phase] += now - ph[8];
            ph[8] = now;
        }
    };
    constexpr int MID_WORDS = 1 << (MID_WL - 6);      // bitmap words of a window; lane l owns the WPL CONSECUTIVE words from l * WPL
    constexpr int WPL = MID_WORDS / WAVE;             // words per lane (2^13 columns: 2, 2^14: 4, 2^16: 16)
    static_assert(MID_WL >= 13 && MID_WL <= 16, "window of the wave-per-row kernel");   // (the owner | column pairs are packed 16 + 16 bits)
    __shared__ unsigned long long bm_s[MID_WAVES][MID_WORDS];
    __shared__ uint16_t sub_s[NUMERIC ? MID_WAVES : 1][NUMERIC ? MID_WORDS : 4];
    __shared__ double acc_s[NUMERIC ? MID_WAVES : 1][NUMERIC ? MID_ACC : 1];
    __shared__ KRec krec_s[MID_WAVES][MID_K];
    __shared__ uint16_t stage_s[NUMERIC ? MID_WAVES : 1][NUMERIC ? MID_ACC : 1];      // columns (offsets in the segment) of one pass, by rank
    __shared__ uint32_t mark_s[MID_WAVES][MID_KEEP * WAVE / 4];      // one-byte marks: which k starts at a position of the current chunk
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    unsigned long long *bm = bm_s[wave];
    uint32_t *bm32 = (uint32_t *)bm;
    uint16_t *sub = sub_s[NUMERIC ? wave : 0];
    double *acc = acc_s[NUMERIC ? wave : 0];
    KRec *krec = krec_s[wave];
    uint16_t *stage = stage_s[NUMERIC ? wave : 0];
    uint32_t *mark32 = mark_s[wave];
    uint8_t *mark8 = (uint8_t *)mark32;
    const bool values = NUMERIC && c_data != nullptr;
    const bool lane_order = (flags & 1u) != 0;
    const bool dbg_no_emit = DEVTOOLS && (flags & 8u), dbg_no_flush = DEVTOOLS && (flags & 16u), dbg_no_add = DEVTOOLS && (flags & 32u);   // timing experiments (option spgemm_debug & 2 / 4 / 8): WRONG results
    uint64_t nwin = (b_cols + (1ull << MID_WL) - 1) >> MID_WL;
    if (nwin == 0) nwin = 1;
#pragma unroll
    for (int i = 0; i < WPL; ++i) bm[lane * WPL + i] = 0;
    wave_sync_lds();
    // The waves draw rows from a counter (the list is sorted by cost, costliest first): which wave takes which row changes
    // nothing in the result — every row is computed by one wave on its own, into its own piece of C.
    // (Written as `for (;;) { q = draw; if (q >= n_mid) break; ... }` this loop was compiled — once the dead timer branches of
    // the developer option were gone from the counting kernel — into a nested loop whose inner level re-entered with the drawn
    // number reset to 0: config 5 never finished on the hardware.  Bisected kernel by kernel in round 3; with the draw in the
    // loop header the same code is compiled as one loop.)
    auto draw_row = [&]() -> uint64_t {
        unsigned int qq = 0;
        if (lane == 0) qq = atomicAdd(next_row, 1u);
        return (uint64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)qq);
    };
    for (uint64_t q = draw_row(); q < n_mid; q = draw_row()) {
        const uint64_t t = mid_list[q];
        const uint64_t r = task_row[t];
        const uint64_t as = (uint64_t)A.indptr[r], ae = (uint64_t)A.indptr[r + 1];
        const uint32_t nk = (uint32_t)(ae - as);                     // <= 64 (row_work_kernel)
        const bool has = lane < nk;
        // positions inside B's row k_j as 32-bit offsets from its start (a row has fewer than 2^32 entries: b_cols < 2^32).
        // Every load here and in the window loop is unconditional (lanes without a k read the row's first k and are masked where
        // lengths are formed): a load under `if (has)` ends in a wait for that load — the "prefetch" of a window edge four windows
        // ahead used to be one more blocking round trip per window.  The edges come from the bucket table alone (this kernel only
        // runs with one: plan_build classes every row as large otherwise): entries of row k before column col, col a multiple of
        // 2048, = bucket[k][min(col / 2048, nb - 1)] — one load, no branch.
        const uint64_t rk = (uint64_t)A.indices[as + (has ? lane : 0u)];
        const uint64_t rs = (uint64_t)B.indptr[rk];
        const uint32_t re = (uint32_t)((uint64_t)B.indptr[rk + 1] - rs);
        double rav = 0.0;
        if (values) rav = A.data[as + (has ? lane : 0u)];
        const uint32_t *__restrict__ bt = B.bucket + rk * B.nb;
        auto edge = [&](uint64_t col) -> uint32_t {          // first entry of my row with column >= col (col a multiple of 2048)
            const uint64_t b = col >> BUCKET_LOG2;
            return bt[b < B.nb - 1 ? b : B.nb - 1];
        };
        uint32_t cur = 0;
        uint32_t q0 = edge(1ull << MID_WL), q1 = edge(2ull << MID_WL), q2 = edge(3ull << MID_WL), q3 = edge(4ull << MID_WL);   // where my row leaves windows w .. w + 3
        uint64_t out = 0;
        if constexpr (NUMERIC) out = off[t];
        uint32_t fresh = 0;
        mark(0);
        const long long t_row = (TIMERS && prof) ? (long long)wall_clock64() : 0;
        // A row with ONE k is a_ik * B_k entry for entry (every sum is 0.0 + a b: the accumulators start at N::zero(), smmp.rs:174-181):
        // streamed, no bitmap, no windows (config 5: 40 000 of the 268 000 rows of this kernel).  Every lane holds that k.
        if (nk == 1) {
            if constexpr (!NUMERIC) {
                fresh = lane == 0 ? re : 0u;
            } else {
                for (uint32_t i0 = 0; i0 < re; i0 += 4 * WAVE) {
                    uint32_t col[4];
                    double bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = i0 + (uint32_t)(u * WAVE) + lane;
                        const uint64_t at = rs + (i < re ? i : 0u);       // (unconditional loads: see load_chunk_n)
                        bv[u] = 0.0;
                        if (values) {                                     // wave-uniform
                            const BRec rec = B.pack[at];
                            col[u] = rec.col;
                            bv[u] = rec.val;
                        } else {
                            col[u] = B.col32[at];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = i0 + (uint32_t)(u * WAVE) + lane;
                        if (i < re) {
                            if (c_indices) c_indices[out + i] = (IDX)col[u];
                            if (values) c_data[out + i] = 0.0 + rav * bv[u];
                        }
                    }
                }
            }
        }
        for (uint64_t w = nk == 1 ? nwin : 0; w < nwin; ++w) {
            const uint64_t win_lo = w << MID_WL;
            const uint32_t win_s = cur, win_e = q0;
            cur = win_e;
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = edge((w + 5) << MID_WL);                            // (past the last window: the whole row)
            const uint32_t winc = wave_incl_scan_u32(has ? win_e - win_s : 0u);
            const uint32_t wtotal = (uint32_t)__builtin_amdgcn_readlane((int)winc, WAVE - 1);
            mark(1);
            if (wtotal == 0) continue;                                // wave-uniform: nothing of the row in this window
            // A window of more entries than one pass of accumulators takes (the dense low columns of a power-law row) is cut into
            // 2, 4 or 8 segments of whole buckets (2048 columns), about 500 entries each or fewer: its outputs then fit one pass
            // nearly always, instead of every pass walking the whole window again (first version: 74 of 152 s of wave time,
            // profiles/r02y).  (A segment of more outputs than accumulators is walked once per pass: rare, bounded by the bucket.)
            uint32_t nseg = 1;
            if (NUMERIC)
                while (nseg < (1u << (MID_WL - BUCKET_LOG2)) && wtotal > nseg * (uint32_t)MID_ACC) nseg <<= 1;
            const uint32_t SEG_COLS = (1u << MID_WL) / nseg;
            uint32_t seg_s = win_s, seg_nxt = edge(win_lo + SEG_COLS);         // (one segment: = win_e)
            for (uint32_t seg = 0; seg < nseg; ++seg) {
            const uint64_t wlo = win_lo + (nseg > 1 ? (uint64_t)seg * SEG_COLS : 0ull);
            const uint32_t ws = seg_s, we = seg + 1 == nseg ? win_e : seg_nxt;
            seg_s = we;
            seg_nxt = edge(wlo + 2 * (uint64_t)SEG_COLS);                  // (asked for one segment ahead; unused after the last)
            const uint32_t len = has ? we - ws : 0u;
            const uint32_t inc = nseg == 1 ? winc : wave_incl_scan_u32(len);
            const uint32_t total = nseg == 1 ? wtotal : (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
            if (total == 0) continue;                                 // wave-uniform
            const uint32_t excl = inc - len;                          // flat position of my k's first entry
            {
                KRec rec;
                rec.base = rs + ws - excl;                            // flat position p of my k lives at entry base + p of B
                rec.a = rav;
                krec[lane] = rec;
            }
            wave_sync_lds();
            const uint32_t nb = (total + WAVE - 1) / WAVE;
            const bool keep = values && nb <= (uint32_t)MID_KEEP;     // wave-uniform
            uint32_t kco[MID_KEEP];                                   // column offset (< 2^16) | owner << 16; 0xFFFFFFFF: no entry
            double kpr[MID_KEEP];
            uint32_t carry = 0;                                       // owner + 1 of the position before the chunk (wave-uniform)
            // One CHUNK = MID_KEEP wave instructions = 256 consecutive positions of the expansion (k ascending, columns ascending
            // inside a k: the reference's order).  Owners: every k that starts inside the chunk leaves its number at its first
            // position (one byte), a max-scan over the positions spreads it over the run — one LDS read and six DPP steps per
            // instruction instead of a search per position.  The loads of the whole chunk are in flight together.
            // (Every load below is UNCONDITIONAL: a lane without an entry reads entry 0 and drops it.  A load under a per-lane
            // condition — `valid ? B[pos] : 0` — is compiled as a branch whose arm ends in a wait for that one load: the eight
            // "independent" loads of a chunk then were eight memory round trips one after the other, in this kernel and in the
            // workgroup kernel, from round 1 on; found in the ISA in round 4.  The instruction count NB is a compile-time number
            // for the same reason: 1, 2, 4 or 8 wave instructions of straight-line code.)
            auto load_chunk_n = [&](uint32_t c0, auto with_value_c, auto nb_c) {
                constexpr bool WITH_VALUE = decltype(with_value_c)::value;
                constexpr int NB = decltype(nb_c)::value;
#pragma unroll
                for (int i = 0; i < MID_KEEP / 4; ++i) mark32[i * WAVE + lane] = 0;
                wave_sync_lds();
                if (len != 0 && excl - c0 < (uint32_t)(MID_KEEP * WAVE)) mark8[excl - c0] = (uint8_t)(lane + 1);
                wave_sync_lds();
                uint64_t pos[NB];
                uint32_t own[NB];
                bool ok[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const uint32_t tpos = c0 + (uint32_t)(b * WAVE) + lane;
                    uint32_t o = wave_incl_max_u32((uint32_t)mark8[b * WAVE + lane]);
                    o = o > carry ? o : carry;
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)o, WAVE - 1);
                    own[b] = o - 1u;                                  // (position 0 always starts a k: o >= 1)
                    ok[b] = tpos < total;
                    const uint64_t at = krec[own[b]].base + tpos;
                    pos[b] = ok[b] ? at : 0ull;
                }
                uint32_t col[NB];
                double bv[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    bv[b] = 0.0;
                    if constexpr (WITH_VALUE) {                       // one 16-byte record per entry
                        const BRec rec = B.pack[pos[b]];
                        col[b] = rec.col;
                        bv[b] = rec.val;
                    } else {
                        col[b] = B.col32[pos[b]];
                    }
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    kco[b] = ok[b] ? (col[b] - (uint32_t)wlo) | (own[b] << 16) : 0xFFFFFFFFu;
                    kpr[b] = 0.0;
                    if constexpr (WITH_VALUE) kpr[b] = krec[own[b]].a * bv[b];
                }
            };
            auto load_chunk = [&](uint32_t c0, auto with_value_c) {
                const uint32_t left = total - c0;
                if (left <= (uint32_t)WAVE) load_chunk_n(c0, with_value_c, std::integral_constant<int, 1>{});
                else if (left <= 2u * WAVE) load_chunk_n(c0, with_value_c, std::integral_constant<int, 2>{});
                else if (left <= 4u * WAVE) load_chunk_n(c0, with_value_c, std::integral_constant<int, 4>{});
                else if (MID_KEEP > 8 && left <= 8u * WAVE) load_chunk_n(c0, with_value_c, std::integral_constant<int, (MID_KEEP > 8 ? 8 : MID_KEEP)>{});
                else load_chunk_n(c0, with_value_c, std::integral_constant<int, MID_KEEP>{});
            };
            auto set_bits = [&](uint32_t c0) {
#pragma unroll
                for (int b = 0; b < MID_KEEP; ++b)
                    if (c0 + (uint32_t)(b * WAVE) < total)            // wave-uniform
                        if (kco[b] != 0xFFFFFFFFu) atomicOr(&bm32[(kco[b] & 0xFFFFu) >> 5], 1u << (kco[b] & 31));
            };
            // ---- bit pass ----
            if (keep) {
                load_chunk(0, std::true_type{});
                set_bits(0);
            } else {
                for (uint32_t c0 = 0; c0 < total; c0 += MID_KEEP * WAVE) {
                    load_chunk(c0, std::false_type{});
                    set_bits(c0);
                }
            }
            wave_sync_lds();
            mark(2);
            // ---- popcount prefix: outputs before each word; a lane's words are consecutive: ONE wave scan ----
            uint32_t run = 0, pre[WPL];
#pragma unroll
            for (int i = 0; i < WPL; ++i) {
                pre[i] = run;
                run += (uint32_t)__popcll(bm[lane * WPL + i]);
            }
            if constexpr (!NUMERIC) {
                fresh += run;
            } else {
                const uint32_t in = wave_incl_scan_u32(run);
                const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)in, WAVE - 1);
#pragma unroll
                for (int i = 0; i < WPL; ++i) sub[lane * WPL + i] = (uint16_t)(in - run + pre[i]);
                wave_sync_lds();
                mark(3);
                // ---- values: passes of MID_ACC outputs; every pass walks the segment's entries and takes its own ----
                // The indices come out sorted for free — the rank of a column IS its place in the row.  Every entry leaves its column
                // at stage[rank] in LDS (several entries of one column: the same value to the same place) and the pass ends with ONE
                // coalesced store of its indices beside the one of its values.  (Stored entry by entry straight to C — 64 scattered
                // 8-byte places per instruction — the kernel wrote 34 GB for 18 GB of C and ran 19 % slower: profiles/r10i.)
                for (uint32_t p0 = 0; (values || c_indices) && p0 < wtot; p0 += MID_ACC) {
                    const uint32_t n_out = wtot - p0 < (uint32_t)MID_ACC ? wtot - p0 : (uint32_t)MID_ACC;
                    for (uint32_t i = lane; values && i < n_out; i += WAVE) acc[i] = 0.0;   // tmp starts at N::zero()
                    wave_sync_lds();
                    auto slot_of = [&](uint32_t cc) -> uint32_t {
                        const uint32_t word = cc >> 6;
                        return (uint32_t)sub[word] + (uint32_t)__popcll(bm[word] & ((1ull << (cc & 63)) - 1ull)) - p0;
                    };
                    auto add_chunk = [&](uint32_t c0) {
#pragma unroll
                        for (int b = 0; b < MID_KEEP; ++b) {
                            if (c0 + (uint32_t)(b * WAVE) < total) {  // wave-uniform
                                const bool kv = kco[b] != 0xFFFFFFFFu;
                                const uint32_t slot = kv ? slot_of(kco[b] & 0xFFFFu) : 0xFFFFFFFFu;
                                if (kv && slot < n_out && c_indices) stage[slot] = (uint16_t)(kco[b] & 0xFFFFu);
                                if (values && !dbg_no_add) add_lanes(kv && slot < n_out, kco[b] >> 16, slot, kpr[b], acc, lane_order);
                            }
                        }
                    };
                    if (keep) {
                        add_chunk(0);
                    } else {
                        carry = 0;
                        for (uint32_t c0 = 0; c0 < total; c0 += MID_KEEP * WAVE) {
                            if (values) load_chunk(c0, std::true_type{});
                            else load_chunk(c0, std::false_type{});
                            add_chunk(c0);
                        }
                    }
                    wave_sync_lds();
                    for (uint32_t i = lane; i < n_out; i += WAVE) {
                        if (c_indices && !dbg_no_emit) c_indices[out + p0 + i] = (IDX)(wlo + (uint64_t)stage[i]);
                        if (values && !dbg_no_flush) c_data[out + p0 + i] = acc[i];
                    }
                    wave_sync_lds();
                }
                out += wtot;
                mark(5);
            }
#pragma unroll
            for (int i = 0; i < WPL; ++i) bm[lane * WPL + i] = 0;
            wave_sync_lds();
            }   // segments of the window
        }
        if constexpr (!NUMERIC) {
            const uint64_t tot = wave_sum_u64(fresh);
            if (lane == 0) count[t] = tot;
        }
        if (TIMERS && prof && lane == 0) {
            const unsigned long long dt = (unsigned long long)((long long)wall_clock64() - t_row);
            const uint64_t u = ub_dbg[r];
            const int c = u < 2048 ? 0 : u < 8192 ? 1 : u < 32768 ? 2 : 3;
            atomicAdd(&prof[8 + 2 * c], dt);
            atomicAdd(&prof[9 + 2 * c], 1ull);
            atomicMax(&prof[16], dt);
        }
    }
    if (TIMERS && prof && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&prof[i], ph[i]);
        atomicMax(&prof[17], (unsigned long long)wall_clock64() - ph[9]);   // longest wave
        atomicAdd(&prof[18], (unsigned long long)wall_clock64() - ph[9]);
        atomicAdd(&prof[19], 1ull);
    }
}

// ranks of the batch's columns (before the turn: only the adds are serialised), then the adds when the token arrives.
// NTOK tokens instead of one: the accumulators are split by the low bits of their rank into NTOK independent sets (the
// order only matters inside one accumulator), each with its own token chain — wave v + 1 adds into set 0 while wave v is
// still busy with set 1: the hand-overs (a completed ds_add, a token write, the next wave's poll) overlap NTOK deep.
__device__ __forceinline__ void batch_add(const Batch &bt, uint32_t U, const unsigned long long *bm, const uint16_t *sub,
                                          const uint32_t *super, uint32_t base_rank, double *acc, uint32_t *token, uint32_t turn,
                                          uint32_t lds_atomic, uint32_t ntok) {
    uint32_t slot[LG_U];
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        const uint32_t word = bt.cc[u] >> 6;
        slot[u] = bt.val[u] ? super[word / SUPER_WORDS] + sub[word] +
                                  (uint32_t)__popcll(bm[word] & ((1ull << (bt.cc[u] & 63)) - 1ull)) - base_rank
                            : 0u;
    }
    // (Finding the run boundaries BEFORE the turn — one cross-lane compare and a ballot per instruction, then scalar bit
    // arithmetic and a predicated add per run inside the turn — was measured slower than the readlane / compare / ballot
    // chain of add_runs: 0.1178 against 0.1151 s, profiles/r03s; 2 and 4 token chains: 0.116 / 0.118 s against 0.115 s.)
    for (uint32_t tk = 0; tk < ntok; ++tk) {
        token_wait(token + tk, turn);
#pragma unroll
        for (int u = 0; u < LG_U; ++u)
            if ((uint32_t)u < U) {
                const bool mine = bt.val[u] && (slot[u] & (ntok - 1)) == tk;
                if (lds_atomic == 2u) add_lanes(mine, bt.own[u], slot[u], bt.pr[u], acc, true);     // one ds_add_f64: the LDS applies the lanes in order
                else add_runs(mine, bt.own[u], slot[u], bt.pr[u], acc, lds_atomic != 0u);
            }
        token_pass(token + tk, turn + 1);       // (a real turn never is 0xFFFFFFFF: the token would have to wrap exactly there; see tok_base)
    }
}

// value walk of one staged group restricted to a pass; returns the number of batches (the token advances by it)
template <int K_CAP>
__device__ __forceinline__ uint32_t walk_values(const BRec *__restrict__ b_pack, uint64_t wlo,
                                                uint32_t gtot, const uint64_t *kS, const uint32_t *kP, const double *kA,
                                                const unsigned long long *bm, const uint16_t *sub, const uint32_t *super,
                                                uint32_t base_rank, double *acc, uint32_t *token, uint32_t tok_base,
                                                uint32_t lds_atomic, uint32_t ntok) {
    const uint32_t wave = threadIdx.x / WAVE;
    const uint32_t U = batch_u(gtot);
    const uint32_t nbatch = (gtot + 64 * U - 1) / (64 * U);
    for (uint32_t b = wave; b < nbatch; b += LG_WAVES) {
        Batch bt;
        batch_load<K_CAP, true>(bt, (const uint32_t *)nullptr, b_pack, wlo, gtot, U, b, kS, kP, kA);
        batch_add(bt, U, bm, sub, super, base_rank, acc, token, tok_base == 0xFFFFFFFFu ? tok_base : tok_base + b, lds_atomic, ntok);
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
                                                              const uint8_t *__restrict__ wlog,
                                                              uint64_t *__restrict__ count,       // symbolic: out
                                                              const uint64_t *__restrict__ off,   // numeric: in
                                                              IDX *__restrict__ c_indices, double *__restrict__ c_data,
                                                              uint32_t xcd_chunk, uint32_t flags,
                                                              unsigned long long *__restrict__ prof,
                                                              const uint64_t *__restrict__ ub_dbg,
                                                              const uint64_t *__restrict__ large_slot,
                                                              unsigned long long *__restrict__ kept_bm, uint64_t kept_words) {
    using Cfg = LgCfg<WL>;
    // phase timers of option spgemm_prof (developer builds only: compiled out of the release library): in LDS (see mid_rows_kernel)
    constexpr bool TIMERS = DEVTOOLS;
    __shared__ unsigned long long ph[10];                               // 0..7 phases, 8 previous mark, 9 task start (100 MHz ticks)
    if (TIMERS && prof && threadIdx.x == 0) {
        for (int i = 0; i < 8; ++i) ph[i] = 0;
        ph[8] = ph[9] = (unsigned long long)wall_clock64();
    }
    auto mark = [&](int phase) {                                        // thread 0's view of the phases
        if (TIMERS && prof && threadIdx.x == 0) {
            const unsigned long long now = (unsigned long long)wall_clock64();
            ph[phase] += now - ph[8];
            ph[8] = now;
        }
    };
    constexpr int WORDS = Cfg::WORDS, WPT = Cfg::WPT, NSUPER = Cfg::NSUPER, ACC_CAP = Cfg::ACC_CAP, K_CAP = Cfg::K_CAP;
    __shared__ unsigned long long bm[WORDS];                 // the window's structure: one bit per column
    __shared__ uint16_t sub[NUMERIC ? WORDS : 1];            // outputs before a word inside its superblock
    __shared__ uint32_t super[NUMERIC ? NSUPER + 1 : 1];     // outputs before each 2048-column superblock
    __shared__ double acc[NUMERIC ? ACC_CAP : 1];            // accumulators of the current pass
    __shared__ uint64_t kS[K_CAP];
    __shared__ uint32_t kP[K_CAP + 1];
    __shared__ double kA[NUMERIC ? K_CAP : 1];
    __shared__ uint64_t wt[16];
    __shared__ uint32_t token[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t t = large_list[task_of_block(blockIdx.x, gridDim.x, xcd_chunk)];
    const uint64_t r = task_row[t];
    const uint64_t jt = t - first_task[r], nt = ntasks[r];
    // The structure of a large row is walked ONCE: the counting kernel leaves the bitmap of every window in global memory
    // (kept_bm: one bit per column of B for every large row, at the slot of the row's first task; plan-owned, bounded), the
    // numeric kernel reads it back — 8 coalesced bytes per 64 columns — instead of walking the row's entries of B a second
    // time for their bits (rounds 1 to 3: 20 % of this kernel and 30 GB of fabric traffic on config 5).
    unsigned long long *__restrict__ gbm = kept_bm ? kept_bm + large_slot[r] * kept_words : nullptr;
    // window width of this row: what row_work_kernel chose (2^winlog, narrower for heavy rows: their tasks ARE windows);
    // the counting kernel is launched in a wider layout (no accumulators) and walks a one-task row in its own, wider windows
    const uint32_t wl = (!NUMERIC && nt == 1) ? (uint32_t)WL : (uint32_t)wlog[r];
    const uint64_t W = 1ull << wl;
    uint64_t nwin = (b_cols + W - 1) >> wl;
    if (nwin == 0) nwin = 1;
    const uint64_t wpt = (nwin + nt - 1) / nt;               // windows per task of this row (row_work_kernel)
    const uint64_t w_begin = jt * wpt;
    uint64_t w_end = w_begin + wpt;
    if (w_end > nwin) w_end = nwin;
    const uint64_t as = (uint64_t)A.indptr[r], ae = (uint64_t)A.indptr[r + 1];
    const bool one_group = ae - as <= (uint64_t)K_CAP;
    const bool values = NUMERIC && c_data != nullptr;
    const uint32_t lds_atomic = (flags & 64u) ? 2u : (flags & 1u);       // how a wave instruction's products are added: 2 one ds_add_f64 (lane order), 1 one per k-run, 0 read-add-write per k-run (A/B)
    const bool no_order = (flags & 4u) != 0;                        // option spgemm_ordered = 0 (supported: same products, unordered atomic adds)
    const bool no_emit = DEVTOOLS && (flags & 8u) != 0;             // timing experiment (option spgemm_debug & 2): WRONG results
    const uint32_t ntok = 1u << ((flags >> 4) & 3u);                          // 1, 2 or 4 token chains (option spgemm_tokens)
    // a row whose k's fit one staged group: thread j keeps k_j, the bounds of B's row k_j and a_ik for the whole task
    const bool mine_k = one_group && tid < (uint32_t)(ae - as) && w_begin < w_end;
    uint64_t rk = 0, rs = 0, re = 0, cur = 0, nxt_e = 0;
    double rav = 0.0;
    if (mine_k) {
        rk = (uint64_t)A.indices[as + tid];
        rs = (uint64_t)B.indptr[rk];
        re = (uint64_t)B.indptr[rk + 1];
        if (values) rav = A.data[as + tid];
        cur = w_begin == 0 ? rs : first_ge(B, rk, rs, rs, re, w_begin << wl);
        nxt_e = w_begin + 1 >= nwin ? re : first_ge(B, rk, rs, cur, re, (w_begin + 1) << wl);
    }
    if (tid < 4) token[tid] = 0;
    for (int i = tid; i < WORDS; i += LG_BLOCK) bm[i] = 0;
    lds_barrier();
    uint64_t out = 0;
    if constexpr (NUMERIC) out = off[t];
    mark(0);
    uint32_t fresh = 0, tok_base = 0;
    (void)tok_base;
    for (uint64_t w = w_begin; w < w_end; ++w) {
        const uint64_t wlo = w << wl, whi = wlo + W;         // whi is not clamped to b_cols: see first_ge
        const uint64_t wcols = (whi < b_cols ? whi : b_cols) - wlo;
        const int words = (int)((wcols + SUPER_WORDS * 64 - 1) / (SUPER_WORDS * 64)) * SUPER_WORDS;   // whole superblocks
        // my k's sub-range in this window; the bound of the next window is requested now and used one window later
        const uint64_t ws = cur, we = nxt_e;
        if (mine_k) {
            cur = we;
            if (w + 1 < w_end) nxt_e = w + 2 >= nwin ? re : first_ge(B, rk, rs, we, re, (w + 2) << wl);
        }
        // ---- bit pass -------------------------------------------------------------------------------
        uint32_t k_total = 0;
        bool any = false;
        if (NUMERIC && gbm) {
            // the bitmap the counting pass kept; the k's of a one-group row are staged all the same (the single-pass walk
            // below uses kS / kP / kA as they are left here), a row of several groups decides by the count of set bits
            for (int i = tid; i < words; i += LG_BLOCK) bm[i] = gbm[(wlo >> 6) + (uint64_t)i];
            if (one_group) {
                k_total = stage_compact<K_CAP>(mine_k ? (uint32_t)(we - ws) : 0u, ws, rav, kS, kP, values ? kA : (double *)nullptr, wt);
                any = k_total != 0;
            } else {
                any = true;
            }
        } else
        for (uint64_t kc = as; kc < ae; kc += K_CAP) {
            const uint32_t n = (ae - kc < (uint64_t)K_CAP) ? (uint32_t)(ae - kc) : (uint32_t)K_CAP;
            const uint32_t gtot =
                one_group ? stage_compact<K_CAP>(mine_k ? (uint32_t)(we - ws) : 0u, ws, rav, kS, kP, values ? kA : (double *)nullptr, wt)
                          : stage_k_group<K_CAP>(A, B, kc, n, wlo, whi, nwin == 1, kS, kP, values ? kA : (double *)nullptr, wt);
            k_total = gtot;
            any |= gtot != 0;
            walk_bits<K_CAP>(B.col32, wlo, gtot, kS, kP, (uint32_t *)bm);
        }
        lds_barrier();
        mark(1);
        if (!any) {                                          // block-uniform; the bitmap is still clear
            if (!NUMERIC && gbm)
                for (int i = tid; i < words; i += LG_BLOCK) gbm[(wlo >> 6) + (uint64_t)i] = 0ull;
            continue;
        }
        if constexpr (!NUMERIC) {
            for (int i = tid; i < words; i += LG_BLOCK) {
                const unsigned long long wbits = bm[i];
                fresh += (uint32_t)__popcll(wbits);
                if (gbm) gbm[(wlo >> 6) + (uint64_t)i] = wbits;
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
            mark(2);
            if (wtot == 0) continue;                               // block-uniform (a kept bitmap without a set bit: it is clear already)
            // indices come out sorted: the set bits of each word in order.  They are STAGED in LDS (the accumulators are idle until
            // the passes start: room for 2 ACC_CAP column offsets at a time) and written out one rank per thread — 512 contiguous
            // bytes per wave instruction.  (Written straight from the bit loops, a store instruction hit 64 scattered 8-byte places;
            // with the stores left out the kernel ran 17 % faster: profiles/r10i.)
            if (c_indices && !no_emit) {                            // (null: C already has its structure)
                uint32_t *stage = (uint32_t *)acc;
                constexpr uint32_t STAGE_CAP = 2u * ACC_CAP;
                for (uint32_t base = 0; base < wtot; base += STAGE_CAP) {      // block-uniform
                    const uint32_t lim = wtot - base < STAGE_CAP ? wtot - base : STAGE_CAP;
#pragma unroll 1
                    for (int i = 0; i < WPT; ++i) {
                        const int word = i * LG_BLOCK + (int)tid;
                        if (word >= words) break;
                        unsigned long long m = bm[word];
                        uint32_t run = super[word / SUPER_WORDS] + sub[word] - base;     // (wraps below the chunk: the compare sorts it out)
                        if (run + (uint32_t)__popcll(m) - 1u >= lim && run >= lim) continue;   // no rank of this word in the chunk
                        while (m) {
                            const int bit = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            if (run < lim) stage[run] = (uint32_t)word * 64u + (uint32_t)bit;
                            ++run;
                        }
                    }
                    lds_barrier();
                    for (uint32_t i = tid; i < lim; i += LG_BLOCK) c_indices[out + base + i] = (IDX)(wlo + (uint64_t)stage[i]);
                    lds_barrier();
                }
            }
            mark(3);
            // ---- values ---------------------------------------------------------------------------------
            // PASSES: the window's superblocks are cut greedily into ranges [pb, pe) of at most ACC_CAP outputs
            // (for a row kept in registers, where my k leaves the NEXT pass is asked for one pass ahead: the bucket-table
            // round trip runs under the walk of the current pass; passes are consecutive, so a pass starts where the
            // previous one ended)
            auto pass_end = [&](uint32_t from) {
                uint32_t to = from + 1;
                while (to < (uint32_t)nsb && super[to + 1] - super[from] <= (uint32_t)ACC_CAP) ++to;
                return to;
            };
            uint32_t pb = 0, pe = values ? pass_end(0) : 0u;
            const bool my_range = one_group && mine_k && we > ws;
            uint64_t pass_s = ws, pass_e = ws;
            if (values && my_range) pass_e = pe == (uint32_t)nsb ? we : first_ge(B, rk, rs, ws, we, wlo + (uint64_t)pe * (SUPER_WORDS * 64));
            while (values && pb < (uint32_t)nsb) {
                const uint32_t pb_next = pe, pe_next = pb_next < (uint32_t)nsb ? pass_end(pb_next) : (uint32_t)nsb;
                uint64_t pass_e_next = pass_e;
                if (my_range && pb_next < (uint32_t)nsb)
                    pass_e_next = pe_next == (uint32_t)nsb ? we : first_ge(B, rk, rs, pass_e, we, wlo + (uint64_t)pe_next * (SUPER_WORDS * 64));
                const uint32_t base_rank = super[pb];
                const uint32_t pass_out = super[pe] - base_rank;
                if (pass_out) {                                  // block-uniform
                    for (uint32_t i = tid; i < pass_out; i += LG_BLOCK) acc[i] = 0.0;   // tmp starts at N::zero()
                    const bool single = pb == 0 && pe == (uint32_t)nsb;
                    const uint64_t plo = wlo + (uint64_t)pb * (SUPER_WORDS * 64);
                    const uint64_t phi = wlo + (uint64_t)pe * (SUPER_WORDS * 64);
                    for (uint64_t kc = as; kc < ae; kc += K_CAP) {
                        const uint32_t n = (ae - kc < (uint64_t)K_CAP) ? (uint32_t)(ae - kc) : (uint32_t)K_CAP;
                        uint32_t gtot;
                        if (single && one_group) {
                            gtot = k_total;                      // kS / kP / kA as the bit pass left them
                            lds_barrier();                       // the accumulators are clear before anybody adds
                        } else if (one_group) {
                            gtot = stage_compact<K_CAP>(my_range ? (uint32_t)(pass_e - pass_s) : 0u, pass_s, rav, kS, kP, kA, wt);
                        } else {
                            gtot = stage_k_group<K_CAP>(A, B, kc, n, single ? wlo : plo, single ? whi : phi, single && nwin == 1,
                                                        kS, kP, kA, wt);
                        }
                        const uint32_t nb_ = walk_values<K_CAP>(B.pack, wlo, gtot, kS, kP, kA, bm, sub, super, base_rank, acc,
                                                                token, no_order ? 0xFFFFFFFFu : tok_base, lds_atomic, ntok);
                        tok_base += nb_;
                    }
                    lds_barrier();                               // every add has been performed
                    mark(4);
                    for (uint32_t i = tid; i < pass_out; i += LG_BLOCK) c_data[out + base_rank + i] = acc[i];
                    lds_barrier();                               // the next pass clears the accumulators
                }
                pass_s = pass_e;
                pass_e = pass_e_next;
                pb = pb_next;
                pe = pe_next;
            }
            out += wtot;
            for (int i = tid; i < words; i += LG_BLOCK) bm[i] = 0;
            lds_barrier();
            mark(5);
        }
    }
    if (TIMERS && prof && tid == 0) {
        prof[blockIdx.x] = (unsigned long long)wall_clock64() - ph[9];
        const double per = (double)ub_dbg[r] / (double)nt;
        const int cls = per < 8192 ? 0 : per < 32768 ? 1 : per < 131072 ? 2 : per < 524288 ? 3 : 4;
        for (int i = 0; i < 6; ++i) atomicAdd(&prof[gridDim.x + cls * 8 + i], ph[i]);
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
        if (p) pool_free(p, cap, dev, true);      // (every kernel of this file runs on the null stream or on the aux stream joined back to it)
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
    uint64_t ntask_total = 0, n_small = 0, n_mid = 0, n_large = 0, n_tiny = 0, c_nnz = 0, nb = 0;
    uint64_t n_micro[3] = {0, 0, 0};   // micro rows of at most 16 / 32 / 64 products and k's (micro_rows_kernel)
    sprs_hip::DevBuf micro_list[3];
    int64_t winlog = 17, midwin = 14;
    uint32_t xcd_chunk = 0;        // how the launch deals the task list to the XCDs (task_of_block)
    uint64_t kept_words = 0;       // 64-bit words per row of the kept bitmaps (0: none kept)
    sprs_hip::DevBuf bcol32, bpack, ent_ext, large_slot, kept_bm, bucket, ub, ntasks, first_task, wlog, task_row, tiny_list, small_list, mid_list, large_list, count, off, counters;
};

namespace sprs_hip {

namespace {

template <typename IDX, typename PTR>
CsrView<IDX, PTR> view_of(const sprs_hip_csmat *m) {
    return CsrView<IDX, PTR>{(const PTR *)m->indptr, (const IDX *)m->indices, m->data, nullptr, 0, nullptr, nullptr};
}

// Option spgemm_overlap = 1: the wave kernels (hash rows, wave-per-row rows) run on a second stream beside the workgroup
// kernel of the large rows.  Measured on config 5: 0.1165 s against 0.1123 s on one stream (profiles/r03e) — the kernels
// are throughput bound and only trade time — so the default is one stream.
struct AuxStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ok = false;
};
AuxStream *aux_stream() {
    static std::mutex mu;
    static std::unordered_map<int, AuxStream> per_device;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    AuxStream &a = per_device[dev];
    if (!a.ok) {
        if (hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        a.ok = true;
    }
    return &a;
}

// ---- the lane-order probe -----------------------------------------------------------------------------------
// add_lanes() relies on a property of the LDS that the ISA manual does not state: the lanes of ONE ds_add_f64 that hit the
// same address are applied in ascending lane order.  Floating-point sums make the order observable, so the library checks it
// once per device before trusting it: 256 waves add 64 values of very different magnitude into 1, 2, 3 or 8 accumulators with
// a single instruction and the host compares the bits with the ascending-lane sums (scripts/probes/lds_add_order.hip is the
// long form: 2.2e7 sums on an MI355X, none different).  A device that fails gets one instruction per k-run (add_runs).
__global__ __launch_bounds__(WAVE) void lane_order_probe_kernel(const double *__restrict__ v, const uint32_t *__restrict__ slot,
                                                                double *__restrict__ out) {
    __shared__ double acc[8];
    const uint32_t lane = threadIdx.x;
    if (lane < 8) acc[lane] = 0.0;
    wave_sync_lds();
    atomicAdd(&acc[slot[blockIdx.x * WAVE + lane]], v[blockIdx.x * WAVE + lane]);
    wave_sync_lds();
    if (lane < 8) out[blockIdx.x * 8 + lane] = acc[lane];
}

bool lds_lane_order_ok() {
    static std::mutex mu;
    static std::unordered_map<int, bool> per_device;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    auto it = per_device.find(dev);
    if (it != per_device.end()) return it->second;
    constexpr int TRIALS = 256;
    std::vector<double> v(TRIALS * WAVE), ref(TRIALS * 8, 0.0), got(TRIALS * 8, -1.0);
    std::vector<uint32_t> slot(TRIALS * WAVE);
    uint64_t s = 0x243F6A8885A308D3ull;
    auto next = [&]() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    static const int NSLOT[4] = {1, 2, 3, 8};
    for (int t = 0; t < TRIALS; ++t)
        for (int l = 0; l < WAVE; ++l) {
            const uint64_t h = next();
            const double m = 0.5 + (double)(h >> 11) * (1.0 / 9007199254740992.0);
            v[t * WAVE + l] = ((h & 1) ? -m : m) * (double)(1ull << (next() % 41)) / 1048576.0;    // magnitudes over 2^-20 .. 2^20
            slot[t * WAVE + l] = (uint32_t)(next() % (uint64_t)NSLOT[t & 3]);
            ref[t * 8 + slot[t * WAVE + l]] += v[t * WAVE + l];                                   // ascending lane order
        }
    bool ok = false;
    DevBuf dv, ds, dout;
    if (dv.alloc(v.size() * 8) == hipSuccess && ds.alloc(slot.size() * 4) == hipSuccess && dout.alloc(got.size() * 8) == hipSuccess &&
        hipMemcpy(dv.p, v.data(), v.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(ds.p, slot.data(), slot.size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(lane_order_probe_kernel, dim3(TRIALS), dim3(WAVE), 0, nullptr, dv.as<double>(), ds.as<uint32_t>(), dout.as<double>());
        if (hipGetLastError() == hipSuccess && hipMemcpy(got.data(), dout.p, got.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            ok = memcmp(got.data(), ref.data(), got.size() * 8) == 0;
    }
    (void)hipGetLastError();
    per_device[dev] = ok;
    return ok;
}

// what the wave kernels are told about the adds (bit 0: one ds_add_f64 per wave instruction)
uint32_t add_flags() { return options().spgemm_lane_order != 2 && lds_lane_order_ok() ? 1u : 0u; }

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
    // the columns of B as 32-bit numbers for the counting walks (4-byte handles already have them)
    if (sizeof(IDX) == 4) {
        B.col32 = (const uint32_t *)b->indices;
    } else {
        SPRS_TRY_HIP(pl->bcol32.alloc((b->nnz ? b->nnz : 1) * sizeof(uint32_t)));
        if (b->nnz) {
            hipLaunchKernelGGL((pack_cols_kernel<IDX>), dim3(2048), dim3(256), 0, stream, B.indices, b->nnz, pl->bcol32.as<uint32_t>());
            SPRS_TRY_HIP(hipGetLastError());
        }
        B.col32 = pl->bcol32.as<uint32_t>();
    }

    DevBuf large_key, mid_key, cls, class_sums, class_totals;
    DevBuf &pos_large = pl->large_slot;      // slot of a large row's bitmap = position of its first task in the (unsorted) large list
    SPRS_TRY_HIP(pl->ub.alloc(rows * 8));
    SPRS_TRY_HIP(pl->wlog.alloc(rows));
    SPRS_TRY_HIP(pl->ntasks.alloc(rows * 8));
    SPRS_TRY_HIP(pl->first_task.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(cls.alloc(rows));
    SPRS_TRY_HIP(pos_large.alloc((rows + 1) * 8));
    // micro rows (lane groups): columns of B below EMPTY, entries of B below 2^40 (the extent word of an entry of A)
    const bool micro = options().spgemm_micro != 2 && b_cols < 0xFFFFFFFFull && (uint64_t)b->nnz <= EXT_START;
    if (micro) SPRS_TRY_HIP(pl->ent_ext.alloc((a->nnz ? a->nnz : 1) * sizeof(uint64_t)));
    // blocks of the class counts (see class_counts_kernel): 2048 rows each, more when that would be more than 65 536 blocks
    uint64_t cls_rb = 2048;
    while ((rows + cls_rb - 1) / cls_rb > 65536) cls_rb *= 2;
    const uint64_t cls_blocks = rows ? (rows + cls_rb - 1) / cls_rb : 1;
    SPRS_TRY_HIP(class_sums.alloc(cls_blocks * CLS_NV * 8));
    SPRS_TRY_HIP(class_totals.alloc(CLS_NV * 8));
    if (rows) {
        uint64_t blocks = (rows + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL((row_work_kernel<IDX, PTR>), dim3((unsigned)blocks), dim3(256), 0, stream, A, B, rows, b_cols,
                           (uint64_t)options().spgemm_heavy, (uint32_t)options().spgemm_winlog, (uint32_t)options().spgemm_minwin,
                           pl->nb ? (uint64_t)options().spgemm_mid : 0ull,      /* the wave-per-row kernel takes its window edges from the bucket table */
                           pl->ub.as<uint64_t>(), pl->ntasks.as<uint64_t>(), cls.as<uint8_t>(), pl->wlog.as<uint8_t>(),
                           micro ? 1u : 0u, micro ? pl->ent_ext.as<uint64_t>() : (uint64_t *)nullptr);
        SPRS_TRY_HIP(hipGetLastError());
        hipLaunchKernelGGL((class_counts_kernel<false>), dim3((unsigned)cls_blocks), dim3(256), 0, stream, (const uint8_t *)cls.as<uint8_t>(),
                           (const uint64_t *)pl->ntasks.as<uint64_t>(), (const uint64_t *)pl->ub.as<uint64_t>(), rows, cls_rb,
                           class_sums.as<uint64_t>(), ClassLists{});
        SPRS_TRY_HIP(hipGetLastError());
    } else {
        SPRS_TRY_HIP(hipMemsetAsync(class_sums.p, 0, CLS_NV * 8, stream));
    }
    hipLaunchKernelGGL(class_sums_kernel, dim3(1), dim3(CLS_NV * WAVE), 0, stream, class_sums.as<uint64_t>(), cls_blocks,
                       class_totals.as<uint64_t>(), pl->first_task.as<uint64_t>(), rows);
    SPRS_TRY_HIP(hipGetLastError());
    uint64_t totals[CLS_NV];
    SPRS_TRY_HIP(hipMemcpy(totals, class_totals.p, sizeof(totals), hipMemcpyDeviceToHost));     // the one read-back that sizes the lists
    pl->ntask_total = totals[0];
    pl->n_large = totals[1];
    pl->n_tiny = totals[2];
    pl->n_small = totals[3];
    pl->n_mid = totals[4];
    for (int m = 0; m < 3; ++m) pl->n_micro[m] = totals[5 + m];
    const uint64_t ntask_total = pl->ntask_total, n_small = pl->n_small, n_mid = pl->n_mid, n_large = pl->n_large, n_tiny = pl->n_tiny;
    for (int m = 0; m < 3; ++m) SPRS_TRY_HIP(pl->micro_list[m].alloc((pl->n_micro[m] ? pl->n_micro[m] : 1) * sizeof(MicroRec)));
    if (n_large > 0x7fffffffull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "too many SpGEMM tasks for one launch");

    SPRS_TRY_HIP(pl->task_row.alloc(ntask_total * 8));
    SPRS_TRY_HIP(pl->tiny_list.alloc(n_tiny * 8));
    SPRS_TRY_HIP(pl->small_list.alloc(n_small * 8));
    SPRS_TRY_HIP(pl->mid_list.alloc(n_mid * 8));
    SPRS_TRY_HIP(pl->large_list.alloc(n_large * 8));
    SPRS_TRY_HIP(large_key.alloc(n_large * 8));
    SPRS_TRY_HIP(mid_key.alloc(n_mid * 8));
    SPRS_TRY_HIP(pl->count.alloc(ntask_total * 8));
    SPRS_TRY_HIP(pl->off.alloc((ntask_total + 1) * 8));
    if (rows) {
        ClassLists lists{pl->task_row.as<uint64_t>(), pl->first_task.as<uint64_t>(), pos_large.as<uint64_t>(), pl->tiny_list.as<uint64_t>(),
                         pl->small_list.as<uint64_t>(), pl->mid_list.as<uint64_t>(), pl->large_list.as<uint64_t>(), large_key.as<uint64_t>(),
                         mid_key.as<uint64_t>(), pl->micro_list[0].as<MicroRec>(), pl->micro_list[1].as<MicroRec>(),
                         pl->micro_list[2].as<MicroRec>()};
        hipLaunchKernelGGL((class_counts_kernel<true>), dim3((unsigned)cls_blocks), dim3(256), 0, stream, (const uint8_t *)cls.as<uint8_t>(),
                           (const uint64_t *)pl->ntasks.as<uint64_t>(), (const uint64_t *)pl->ub.as<uint64_t>(), rows, cls_rb,
                           class_sums.as<uint64_t>(), lists);
        SPRS_TRY_HIP(hipGetLastError());
    }
    // costliest tasks first (stable sort by cost class); option spgemm_task_order = 2 keeps the row order (A/B)
    if (n_large > 1 && options().spgemm_task_order != 2)
        SPRS_TRY(radix_sort_pairs(large_key.as<uint64_t>(), pl->large_list.as<uint64_t>(), n_large, {{0, 6}}, stream));
    if (n_mid > 1 && options().spgemm_task_order != 2)
        SPRS_TRY(radix_sort_pairs(mid_key.as<uint64_t>(), pl->mid_list.as<uint64_t>(), n_mid, {{0, 6}}, stream));

    // large rows first on the main stream (the long tasks start at once), the wave kernels beside them on the second stream
    AuxStream *aux = options().spgemm_overlap && !options().spgemm_prof ? aux_stream() : nullptr;
    hipStream_t wstream = aux ? aux->s : stream;
    if (aux) {
        SPRS_TRY_HIP(hipEventRecord(aux->fork, stream));
        SPRS_TRY_HIP(hipStreamWaitEvent(aux->s, aux->fork, 0));
    }
    // ONE launch for all large tasks, in the LDS layout of the window width (option spgemm_winlog)
    if (n_large) {
        const dim3 g((unsigned)n_large), blk(LG_BLOCK);
#define SPRS_LG_SYM(WL)                                                                                              \
    hipLaunchKernelGGL((large_rows_kernel<WL, IDX, PTR, false, 1>), g, blk, 0, stream, A, B, b_cols,                    \
                       pl->large_list.as<uint64_t>(), pl->task_row.as<uint64_t>(), pl->first_task.as<uint64_t>(),    \
                       pl->ntasks.as<uint64_t>(), (const uint8_t *)pl->wlog.as<uint8_t>(), pl->count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, \
                       (double *)nullptr, pl->xcd_chunk, 0u, (unsigned long long *)nullptr, (const uint64_t *)nullptr,       \
                       pl->large_slot.as<uint64_t>(), pl->kept_words ? pl->kept_bm.as<unsigned long long>() : (unsigned long long *)nullptr, pl->kept_words)
        // bitmaps of the large rows, kept for the numeric phase (one bit per column per large task slot; config 5: 3.8 GB):
        // bounded by 16 GiB and by a quarter of the free device memory, otherwise the numeric kernel walks for its bits
        pl->kept_words = 0;
        if (options().spgemm_keep_bits) {
            const uint64_t words_row = ((b_cols + SUPER_WORDS * 64 - 1) / (SUPER_WORDS * 64)) * SUPER_WORDS;
            const uint64_t bytes = n_large * words_row * 8;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes <= (16ull << 30) && bytes <= free_b / 4 &&
                pl->kept_bm.alloc(bytes) == hipSuccess) {
                pl->kept_words = words_row;
            } else {
                (void)hipGetLastError();
                clear_error();
            }
        }
        switch (pl->winlog) {                    // (counting: one layout wider than the numeric kernel's)
            case 16: SPRS_LG_SYM(17); break;
            case 18: SPRS_LG_SYM(19); break;
            case 19: SPRS_LG_SYM(19); break;
            default: SPRS_LG_SYM(18); break;
        }
#undef SPRS_LG_SYM
        SPRS_TRY_HIP(hipGetLastError());
    }
    auto small_grid = [&](uint64_t n_tasks) {
        uint64_t g = (n_tasks + SM_WAVES - 1) / SM_WAVES;
        if (g > 256 * 32) g = 256 * 32;
        return dim3((unsigned)g);
    };
    auto micro_grid = [&](uint64_t n_rows, int per_wave) {
        uint64_t g = ((n_rows + per_wave - 1) / per_wave + 3) / 4;       // four waves per block
        if (g > 256 * 8) g = 256 * 8;                                    // what the chip holds at once; the waves stride over the list
        return dim3((unsigned)g);
    };
#define SPRS_MICRO_SYM(M, GV, KV)                                                                                                     \
    if (pl->n_micro[M]) {                                                                                                           \
        hipLaunchKernelGGL((micro_rows_kernel<IDX, PTR, false, GV, KV>), micro_grid(pl->n_micro[M], WAVE / GV), dim3(256), 0, wstream, A, B, (const uint64_t *)pl->ent_ext.as<uint64_t>(), \
                           (const MicroRec *)pl->micro_list[M].as<MicroRec>(), pl->n_micro[M], pl->count.as<uint64_t>(),             \
                           (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr);                                            \
        SPRS_TRY_HIP(hipGetLastError());                                                                                            \
    }
    if (b_cols <= MICRO_KEY32_COLS) {                                   // (column, position) sort keys of one word
        SPRS_MICRO_SYM(0, 16, uint32_t)
        SPRS_MICRO_SYM(1, 32, uint32_t)
        SPRS_MICRO_SYM(2, 64, uint32_t)
    } else {
        SPRS_MICRO_SYM(0, 16, uint64_t)
        SPRS_MICRO_SYM(1, 32, uint64_t)
        SPRS_MICRO_SYM(2, 64, uint64_t)
    }
#undef SPRS_MICRO_SYM
    if (n_tiny) {
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, false, TINY_TAB>), small_grid(n_tiny), dim3(SM_BLOCK), 0, wstream,
                           A, B, pl->tiny_list.as<uint64_t>(), n_tiny, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr, 0u, 0u);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (n_small) {
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, false, SMALL_TAB>), small_grid(n_small), dim3(SM_BLOCK), 0, wstream,
                           A, B, pl->small_list.as<uint64_t>(), n_small, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr, 0u, 0u);
        SPRS_TRY_HIP(hipGetLastError());
    }
    pl->midwin = options().spgemm_midwin;
    auto mid_grid = [&](uint64_t n_tasks) {
        uint64_t g = (n_tasks + MID_WAVES - 1) / MID_WAVES;
        if (g > 256 * 12) g = 256 * 12;          // what fits the CUs at once; the waves stride over the (cost-sorted) list
        return dim3((unsigned)g);
    };
    if (n_mid) {
#define SPRS_MID_SYM(WL, KP)                                                                                             \
    hipLaunchKernelGGL((mid_rows_kernel<IDX, PTR, false, WL, KP>), mid_grid(n_mid), dim3(MID_BLOCK), 0, wstream, A, B, b_cols, \
                       pl->mid_list.as<uint64_t>(), n_mid, pl->task_row.as<uint64_t>(), pl->count.as<uint64_t>(),    \
                       (const uint64_t *)nullptr, (IDX *)nullptr, (double *)nullptr, (unsigned long long *)nullptr, (const uint64_t *)nullptr, \
                       pl->counters.as<unsigned int>(), 0u)
        SPRS_TRY_HIP(pl->counters.alloc(64));
        SPRS_TRY_HIP(hipMemsetAsync(pl->counters.p, 0, 64, wstream));
        // the counting kernel has no accumulators: windows of 2^16 columns (four times fewer window prologues per row)
        const bool k16 = options().spgemm_mid_keep_sym >= 16;
        if (options().spgemm_midwin_sym >= 15) { if (k16) SPRS_MID_SYM(16, 16); else SPRS_MID_SYM(16, 8); }
        else { if (k16) SPRS_MID_SYM(14, 16); else SPRS_MID_SYM(14, 8); }
#undef SPRS_MID_SYM
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (aux) {
        SPRS_TRY_HIP(hipEventRecord(aux->join, aux->s));
        SPRS_TRY_HIP(hipStreamWaitEvent(stream, aux->join, 0));
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
    B.col32 = sizeof(IDX) == 4 ? (const uint32_t *)b->indices : pl->bcol32.as<uint32_t>();
    if (values && (pl->n_mid || pl->n_large)) {
        // {column, value} records of B for the value walks; rebuilt per call: the values of b may have changed since the plan was made
        SPRS_TRY_HIP(pl->bpack.alloc((b->nnz ? b->nnz : 1) * sizeof(BRec)));
        if (b->nnz) {
            hipLaunchKernelGGL((pack_entries_kernel<IDX>), dim3(2048), dim3(256), 0, stream, B.indices, B.data, b->nnz, pl->bpack.as<BRec>());
            SPRS_TRY_HIP(hipGetLastError());
        }
        B.pack = pl->bpack.as<BRec>();
    }
    const uint64_t n_tiny = pl->n_tiny, n_small = pl->n_small, n_mid = pl->n_mid, n_large = pl->n_large;
    double *c_values = values ? c->data : nullptr;
    IDX *c_indices = indices ? (IDX *)c->indices : nullptr;
    auto small_grid = [&](uint64_t n_tasks) {
        uint64_t g = (n_tasks + SM_WAVES - 1) / SM_WAVES;
        if (g > 256 * 32) g = 256 * 32;
        return dim3((unsigned)g);
    };
    AuxStream *aux = options().spgemm_overlap && !options().spgemm_prof ? aux_stream() : nullptr;
    hipStream_t wstream = aux ? aux->s : stream;
    if (aux) {
        SPRS_TRY_HIP(hipEventRecord(aux->fork, stream));
        SPRS_TRY_HIP(hipStreamWaitEvent(aux->s, aux->fork, 0));
    }
    if (n_large) {
        const dim3 g((unsigned)n_large), blk(LG_BLOCK);
        // bit 2: the adds of a workgroup's waves are NOT put in the reference's order (option spgemm_ordered = 0: every C(i,j) still
        // is the sum of the same products, added by LDS atomics in whatever order the waves arrive — rounding-level differences,
        // not reproducible run to run; structure unaffected).  Needs the atomic form of the add.
        const bool unordered = options().spgemm_ordered == 0;
        const uint32_t flags = ((options().spgemm_lds_atomic || unordered) ? 1u : 0u) |                                (unordered ? 4u : 0u) | ((uint32_t)(options().spgemm_debug & 3) << 2) |
                               ((options().spgemm_tokens >= 4 ? 2u : options().spgemm_tokens >= 2 ? 1u : 0u) << 4) |
                               ((options().spgemm_lds_atomic || unordered) && add_flags() ? 64u : 0u);
#define SPRS_LG_NUM(WL, OCC)                                                                                         \
    hipLaunchKernelGGL((large_rows_kernel<WL, IDX, PTR, true, OCC>), g, blk, 0, stream, A, B, pl->b_cols,            \
                       pl->large_list.as<uint64_t>(), pl->task_row.as<uint64_t>(), pl->first_task.as<uint64_t>(),    \
                       pl->ntasks.as<uint64_t>(), (const uint8_t *)pl->wlog.as<uint8_t>(), pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), c_indices, \
                       c_values, pl->xcd_chunk, flags, prof.as<unsigned long long>(), pl->ub.as<uint64_t>(),                      \
                       pl->large_slot.as<uint64_t>(), pl->kept_words ? pl->kept_bm.as<unsigned long long>() : (unsigned long long *)nullptr, pl->kept_words)
        DevBuf prof;
        if (DEVTOOLS && options().spgemm_prof) {
            SPRS_TRY_HIP(prof.alloc((n_large + 40) * 8));
            SPRS_TRY_HIP(hipMemsetAsync(prof.p, 0, (n_large + 40) * 8, stream));
        }
        const bool occ3 = options().spgemm_occupancy != 2;
        switch (pl->winlog) {
            case 16: if (occ3) SPRS_LG_NUM(16, 6); else SPRS_LG_NUM(16, 4); break;
            case 18: SPRS_LG_NUM(18, 1); break;
            case 19: SPRS_LG_NUM(19, 1); break;
            default: if (occ3) SPRS_LG_NUM(17, 6); else SPRS_LG_NUM(17, 4); break;
        }
#undef SPRS_LG_NUM
        if (DEVTOOLS && prof.p) {
            // debug: 100 MHz ticks per large task, by block (= position in the launch order)
            SPRS_TRY_HIP(hipStreamSynchronize(stream));
            unsigned long long phs[40];
            (void)hipMemcpy(phs, (char *)prof.p + n_large * 8, 320, hipMemcpyDeviceToHost);
            for (int c = 0; c < 5; ++c)
                fprintf(stderr, "[spgemm_prof] class %d, thread-0 time by phase (ms of workgroup time): prologue %.1f, stage+bits %.1f, prefix %.1f, "
                                "emit %.1f, values %.1f, flush+clear %.1f\n", c, phs[c * 8 + 0] / 1e5, phs[c * 8 + 1] / 1e5, phs[c * 8 + 2] / 1e5,
                        phs[c * 8 + 3] / 1e5, phs[c * 8 + 4] / 1e5, phs[c * 8 + 5] / 1e5);
            std::vector<unsigned long long> tk(n_large), lst(n_large), trow(pl->ntask_total), ubv(pl->rows), ntk(pl->rows);
            (void)hipMemcpy(tk.data(), prof.p, n_large * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(lst.data(), pl->large_list.p, n_large * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(trow.data(), pl->task_row.p, pl->ntask_total * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(ubv.data(), pl->ub.p, pl->rows * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(ntk.data(), pl->ntasks.p, pl->rows * 8, hipMemcpyDeviceToHost);
            std::vector<uint64_t> a_ip(pl->rows + 1);
            if (sizeof(PTR) == 8) (void)hipMemcpy(a_ip.data(), a->indptr, (pl->rows + 1) * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            unsigned long long mx = 0;
            // classes by products per task: < 2^13, < 2^15, < 2^17, < 2^19, rest
            double csum[5] = {0, 0, 0, 0, 0}, cprod[5] = {0, 0, 0, 0, 0};
            unsigned long long cn[5] = {0, 0, 0, 0, 0};
            std::vector<std::pair<unsigned long long, uint64_t>> top;
            for (uint64_t i = 0; i < n_large; ++i) {
                const uint64_t r = trow[lst[i]];
                const double per = (double)ubv[r] / (double)ntk[r];
                const int c = per < 8192 ? 0 : per < 32768 ? 1 : per < 131072 ? 2 : per < 524288 ? 3 : 4;
                csum[c] += (double)tk[i];
                cprod[c] += per;
                ++cn[c];
                sum += (double)tk[i];
                if (tk[i] > mx) mx = tk[i];
                top.push_back({tk[i], i});
            }
            std::partial_sort(top.begin(), top.begin() + (top.size() < 8 ? top.size() : 8), top.end(),
                              [](const auto &x, const auto &y) { return x.first > y.first; });
            fprintf(stderr, "[spgemm_prof] large tasks %llu: sum %.1f ms of workgroup time, longest %.3f ms\n",
                    (unsigned long long)n_large, sum / 1e5, (double)mx / 1e5);
            for (int c = 0; c < 5; ++c)
                fprintf(stderr, "[spgemm_prof]   class %d: %llu tasks, %.3e products, %.1f ms of workgroup time (%.2f us per task, %.2f ns per product)\n",
                        c, cn[c], cprod[c], csum[c] / 1e5, cn[c] ? csum[c] / 1e2 / (double)cn[c] : 0.0, cprod[c] ? csum[c] * 10.0 / cprod[c] : 0.0);
            for (size_t j = 0; j < top.size() && j < 8; ++j) {
                const uint64_t i = top[j].second, r = trow[lst[i]];
                fprintf(stderr, "[spgemm_prof]   top %zu: block %llu row %llu products %llu tasks of the row %llu k's %llu: %.3f ms\n", j,
                        (unsigned long long)i, (unsigned long long)r, ubv[r], ntk[r],
                        sizeof(PTR) == 8 ? (unsigned long long)(a_ip[r + 1] - a_ip[r]) : 0ull, (double)top[j].first / 1e5);
            }
        }
    }
    // bins of the hash kernels' rank pass: the column space cut into SM_NBIN equal power-of-two ranges
    uint32_t bin_shift = 0;
    while (bin_shift < 32 && ((pl->b_cols - (pl->b_cols ? 1 : 0)) >> bin_shift) >= (uint64_t)SM_NBIN) ++bin_shift;
    const uint32_t small_flags = add_flags();
#define SPRS_MICRO_NUM(M, GV, KV)                                                                                                     \
    if (pl->n_micro[M]) {                                                                                                           \
        uint64_t mg = ((pl->n_micro[M] + (WAVE / GV) - 1) / (WAVE / GV) + 3) / 4;                                                     \
        if (mg > 256 * 8) mg = 256 * 8;                                                                                             \
        hipLaunchKernelGGL((micro_rows_kernel<IDX, PTR, true, GV, KV>), dim3((unsigned)mg), dim3(256), 0, wstream, A, B, (const uint64_t *)pl->ent_ext.as<uint64_t>(), \
                           (const MicroRec *)pl->micro_list[M].as<MicroRec>(), pl->n_micro[M], pl->count.as<uint64_t>(),             \
                           (const uint64_t *)pl->off.as<uint64_t>(), c_indices, c_values);                                           \
        SPRS_TRY_HIP(hipGetLastError());                                                                                            \
    }
    if (pl->b_cols <= MICRO_KEY32_COLS) {                                   // (column, position) sort keys of one word
        SPRS_MICRO_NUM(0, 16, uint32_t)
        SPRS_MICRO_NUM(1, 32, uint32_t)
        SPRS_MICRO_NUM(2, 64, uint32_t)
    } else {
        SPRS_MICRO_NUM(0, 16, uint64_t)
        SPRS_MICRO_NUM(1, 32, uint64_t)
        SPRS_MICRO_NUM(2, 64, uint64_t)
    }
#undef SPRS_MICRO_NUM
    if (n_tiny)
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, true, TINY_TAB>), small_grid(n_tiny), dim3(SM_BLOCK), 0, wstream, A,
                           B, pl->tiny_list.as<uint64_t>(), n_tiny, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), c_indices, c_values, bin_shift, small_flags);
    if (n_small)
        hipLaunchKernelGGL((small_rows_kernel<IDX, PTR, true, SMALL_TAB>), small_grid(n_small), dim3(SM_BLOCK), 0, wstream,
                           A, B, pl->small_list.as<uint64_t>(), n_small, pl->task_row.as<uint64_t>(), pl->ub.as<uint64_t>(),
                           pl->count.as<uint64_t>(), pl->off.as<uint64_t>(), c_indices, c_values, bin_shift, small_flags);
    if (n_mid) {
        uint64_t g = (n_mid + MID_WAVES - 1) / MID_WAVES;
        if (g > 256 * 12) g = 256 * 12;
        DevBuf mprof;
        if (DEVTOOLS && options().spgemm_prof) {
            SPRS_TRY_HIP(mprof.alloc(256));
            SPRS_TRY_HIP(hipMemsetAsync(mprof.p, 0, 256, stream));
        }
#define SPRS_MID_NUM(WL, KP)                                                                                             \
    hipLaunchKernelGGL((mid_rows_kernel<IDX, PTR, true, WL, KP>), dim3((unsigned)g), dim3(MID_BLOCK), 0, wstream, A, B, pl->b_cols, \
                       pl->mid_list.as<uint64_t>(), n_mid, pl->task_row.as<uint64_t>(), pl->count.as<uint64_t>(),    \
                       pl->off.as<uint64_t>(), c_indices, c_values, mprof.as<unsigned long long>(), pl->ub.as<uint64_t>(),   \
                       pl->counters.as<unsigned int>() + 4, wave_flags)
        SPRS_TRY_HIP(hipMemsetAsync(pl->counters.as<unsigned int>() + 4, 0, 4, wstream));
        const uint32_t wave_flags = add_flags() | (DEVTOOLS ? (uint32_t)options().spgemm_debug << 2 : 0u);
        const bool k8 = options().spgemm_mid_keep >= 8;
        if (pl->midwin >= 15) { if (k8) SPRS_MID_NUM(15, 8); else SPRS_MID_NUM(15, 4); }
        else { if (k8) SPRS_MID_NUM(14, 8); else SPRS_MID_NUM(14, 4); }
#undef SPRS_MID_NUM
        if (DEVTOOLS && mprof.p) {
            unsigned long long h[32];
            SPRS_TRY_HIP(hipStreamSynchronize(stream));
            (void)hipMemcpy(h, mprof.p, 256, hipMemcpyDeviceToHost);
            fprintf(stderr, "[spgemm_prof] mid rows by products (< 2048, < 8192, < 32768, rest): %llu rows %.1f ms | %llu rows %.1f ms | %llu rows %.1f ms | "
                            "%llu rows %.1f ms of wave time; longest row %.3f ms; waves %llu, mean wave %.3f ms, longest wave %.3f ms\n",
                    h[9], h[8] / 1e5, h[11], h[10] / 1e5, h[13], h[12] / 1e5, h[15], h[14] / 1e5, h[16] / 1e5, h[19],
                    h[19] ? h[18] / 1e5 / (double)h[19] : 0.0, h[17] / 1e5);
            fprintf(stderr, "[spgemm_prof] mid rows %llu, lane-0 time by phase (ms of wave time): row prologue %.1f, bounds+scan %.1f, stage+loads+bits %.1f, "
                            "prefix %.1f, emit %.1f, values+flush %.1f\n", (unsigned long long)n_mid, h[0] / 1e5, h[1] / 1e5, h[2] / 1e5, h[3] / 1e5,
                    h[4] / 1e5, h[5] / 1e5);
        }
    }
    if (aux) {
        SPRS_TRY_HIP(hipEventRecord(aux->join, aux->s));
        SPRS_TRY_HIP(hipStreamWaitEvent(stream, aux->join, 0));
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
