// BANDED SpMV PLAN for power-law matrices whose x does not fit the L2s (BASELINE config 4, R-MAT 10M)
// — device twin of prod::mul_acc_mat_vec_csr (sprs/src/sparse/prod.rs:103-127) and of the one-column
// prod::csr_mulacc_dense_colmaj (prod.rs:274-298), like spmv.hip.
//
// Why: with every x[col] gathered through L1/L2 the kernel moves one 128-byte line from L2 to L1 per
// 8 useful bytes (0.87 L2 requests per non-zero on R-MAT 10M, profiles/r01zr...): it is bound by the
// L2 -> L1 fill bandwidth, not by HBM, and stops at 38 % of the HBM roofline.  On a power-law matrix a
// small set of columns carries most of the entries (R-MAT 10M: the 2 M most referenced columns of 1e7
// hold 96.7 % of the non-zeros), and the plan may lay the matrix out as it likes.  So:
//
//   * columns are relabelled by popularity class (rl_* kernels, spmv_shared.hpp), x is permuted into
//     that order once per SpMV (xp);
//   * HOT BAND: labels [0, nh * XT), XT = 16384 (or 8192).  Hot slice k = the entries of the long rows
//     (>= split entries) whose label lies in [XT k, XT (k+1)), row after row: 8-byte value + 16-BIT local
//     column id (10 B per entry instead of 16).  The hot kernel keeps the XT x entries of a slice in LDS
//     (128 KiB, loaded coalesced) and gathers from LDS: no L1/L2 traffic per entry at all, the slice
//     streams at HBM speed.  Round 2 used 8192-label tiles (64 KiB): on R-MAT 10M 128 such slices cover
//     1 M labels with 45 M (row, slice) pairs; 128 slices of 16384 cover 2 M labels (cold entries of the
//     long rows 25.8 M -> 9.9 M) with 40 M pairs — fewer gathers AND fewer partial sums;
//   * COLD REST: the other entries of the long rows, in 8 pieces by a hash of their x line (piece s runs
//     on XCD s), optionally in several label ranges ("phases");
//   * SHORT ROWS: one piece over the rows that are not empty, whole; it writes y directly.
//   A piece writes one partial sum per (row, piece) pair it has, COMPACTLY, in the order of its own row
//   list; band_reduce_kernel adds the partials of a long row in piece order, finding them through
//   per-(64 rows, piece) presence masks and base offsets.  No float atomics: bit-reproducible run to run.
// The kernels and the wave-tile layout: spmv_band_kernels.hpp.
#include "spmv_band_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace sprs_hip {

namespace {

constexpr int MAX_HOT = 384;
constexpr int MAX_PHASES = 8;
constexpr int MAX_PIECES = MAX_HOT + 8 * MAX_PHASES + 1;

__global__ __launch_bounds__(256) void bp_inverse_kernel(const uint32_t *__restrict__ perm, uint64_t cols, uint32_t hot_labels,
                                                         uint32_t *__restrict__ inv_hot) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    const uint32_t l = perm[j];
    if (l < hot_labels) inv_hot[l] = (uint32_t)j;
}

// ---------------------------------------------------------------------------------------------
// plan building (one-time, on the device)
// ---------------------------------------------------------------------------------------------
struct SliceMap {       // label -> piece
    uint32_t nh, phases, xt_log2;
    uint64_t hot_labels;      // nh << xt_log2
    uint64_t phase_width;     // labels per phase of the cold rest
};

__device__ __forceinline__ uint32_t piece_of_label(const SliceMap &m, uint64_t label) {
    if (label < m.hot_labels) return (uint32_t)(label >> m.xt_log2);
    uint64_t ph = (label - m.hot_labels) / m.phase_width;
    if (ph >= m.phases) ph = m.phases - 1;
    return m.nh + (uint32_t)ph * 8u + x_slice(label);
}

template <typename PTR>
__global__ void bp_classify_kernel(const PTR *__restrict__ indptr, uint64_t rows, uint64_t split,
                                   uint64_t *__restrict__ short_flag, uint64_t *__restrict__ short_len,
                                   uint64_t *__restrict__ long_flag) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t len = (uint64_t)indptr[r + 1] - (uint64_t)indptr[r];
    const bool is_long = len >= split;
    short_flag[r] = (!is_long && len) ? 1 : 0;
    short_len[r] = is_long ? 0 : len;
    long_flag[r] = is_long ? 1 : 0;
}

// short rows -> their compact CSR piece (labels instead of columns); long rows -> long_rows.
// BUCKETS (round 6): a range of the short piece is `R` entries (cold_tiles wave tiles) walked by one wave; a row that runs on
// from one range into the next needs a fix-up launch behind the short rows (band_carry_kernel<true>).  Short rows have
// fewer than `split` entries, so the piece is laid out with gaps instead: the rows whose position d in the gap-free
// concatenation lies in [k R', (k + 1) R'), R' = R - (split - 1), go to range k, packed from its first entry on — they hold
// at most R' + split - 1 = R entries — and the rest of the range is padding (value 0, label `cols`: xp[cols] is always 0.0).
// Every range then begins with a row start: no heads, no records, no launch; the price is (split - 1) / R of the piece
// (R-MAT 1M: 7 / 512 of 0.9 M entries).  bucket0[k] = position of the first row of bucket k in the gap-free concatenation.
struct ShortBuckets {
    const uint64_t *bucket0;     // null: the gap-free layout (rows long enough to make the gaps expensive)
    uint64_t Rp, R;
};

__device__ __forceinline__ uint64_t short_place(const ShortBuckets &sb, uint64_t d) {
    if (!sb.bucket0) return d;
    const uint64_t k = d / sb.Rp;
    return k * sb.R + (d - sb.bucket0[k]);
}

__global__ void bp_bucket_starts_kernel(const uint64_t *__restrict__ short_ptr, uint64_t rows, uint64_t Rp, uint64_t nb,
                                        uint64_t *__restrict__ bucket0) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    const uint64_t target = k * Rp;
    uint64_t lo = 0, hi = rows;                       // first row r with short_ptr[r] >= target (short_ptr[rows] = all short entries > target)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (short_ptr[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    bucket0[k] = short_ptr[lo];
}

// Number of a long row in the plan's row lists.  The reduction takes the long rows in blocks of 64 (one wave each).  In natural
// order the first block of a power-law matrix is 64 hub rows — on R-MAT 1M 266 of the 27 000 heads of rows that run on from one
// range into the next, and most (row, slice) pairs — and its wave is the reduction's critical path.  The first 61 blocks' rows
// are therefore dealt round-robin (61: the hubs of an R-MAT matrix sit at the powers of two and their sums; a power-of-two
// stride would put them into one block again): row j < 64 * B goes to lane j / B of block j % B.  Every table of the plan is
// built from this numbering (long_rows[j] is the row), so nothing else knows about it.
__device__ __forceinline__ uint64_t long_row_number(uint64_t j, uint64_t n_long) {
    const uint64_t full = n_long / WAVE;
    const uint64_t B = full < 61 ? full : 61;
    if (B < 2 || j >= B * WAVE) return j;
    return (j % B) * WAVE + j / B;
}

// mode 0: the row lists (s_rowidx, long_rows); mode 1: the entries in the wave-tile layout of band_cold_kernel and the
// compact rows' positions (s_ptr), both in the layout `sb` describes
template <typename IDX, typename PTR>
__global__ void bp_fill_short_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                     const double *__restrict__ data, uint64_t rows,
                                     const uint64_t *__restrict__ short_pos, const uint64_t *__restrict__ short_ptr,
                                     const uint64_t *__restrict__ long_pos, const uint32_t *__restrict__ perm,
                                     uint32_t *__restrict__ s_rowidx, uint32_t *__restrict__ s_ptr,
                                     uint32_t *__restrict__ s_cid, double *__restrict__ s_val,
                                     uint32_t *__restrict__ long_rows, uint64_t split, int mode, ShortBuckets sb, uint64_t padded_total,
                                     uint64_t n_long) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    if (r == rows) {
        if (mode == 1) s_ptr[short_pos[rows]] = (uint32_t)padded_total;
        return;
    }
    const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
    if (e == s) return;
    if (e - s >= split) {
        if (mode == 0) long_rows[long_row_number(long_pos[r], n_long)] = (uint32_t)r;
        return;
    }
    const uint64_t i = short_pos[r];
    if (mode == 0) {
        s_rowidx[i] = (uint32_t)r;
        return;
    }
    uint64_t d = short_place(sb, short_ptr[r]);
    s_ptr[i] = (uint32_t)d;
    for (uint64_t p = s; p < e; ++p, ++d) {
        const uint64_t tile = d / WT;
        const uint32_t t = (uint32_t)(d % WT);
        const uint32_t ll = t / EPL, q = t % EPL;
        const uint32_t label = perm[indices[p]];
        s_cid[tile * WT + (q / 4) * (WAVE * 4) + ll * 4 + (q & 3u)] = label | (p == s ? ROW_START32 : 0u);
        s_val[tile * WT + (q / 2) * (WAVE * 2) + ll * 2 + (q & 1u)] = data[p];
    }
}

// entries of long row j per piece: cnt[k * n_long + j] (and 1 where that is not zero)
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void bp_count_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                       const uint32_t *__restrict__ long_rows, uint64_t n_long,
                                                       const uint32_t *__restrict__ perm, SliceMap map, uint32_t npieces,
                                                       uint64_t *__restrict__ cnt, uint64_t *__restrict__ nz) {
    __shared__ uint32_t hist[4][MAX_PIECES];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    for (uint64_t j = w0; j < n_long; j += nw) {
        for (uint32_t k = lane; k < npieces; k += WAVE) hist[wave][k] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint64_t r = long_rows[j];
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        for (uint64_t p = s + lane; p < e; p += WAVE) atomicAdd(&hist[wave][piece_of_label(map, perm[indices[p]])], 1u);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k < npieces; k += WAVE) {
            const uint32_t c = hist[wave][k];
            cnt[(uint64_t)k * n_long + j] = c;
            nz[(uint64_t)k * n_long + j] = c ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void bp_starts_kernel(const uint64_t *__restrict__ pos, const uint64_t *__restrict__ pair, uint64_t n_long,
                                 uint32_t npieces, uint64_t *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > npieces) return;
    out[k] = pos[(uint64_t)k * n_long];
    out[npieces + 1 + k] = pair[(uint64_t)k * n_long];
}

struct PieceBuild {          // host-computed placement of a piece, read by the scatter kernels
    uint64_t start;          // position of the piece in the piece-major concatenation (scan of cnt)
    uint64_t pair0;          // first (row, piece) pair of the piece (scan of nz)
    uint64_t ent0;           // first entry in the class's arrays
    uint64_t nnz;
    uint32_t ptr_off, row_off;   // offsets into ptr_all / rowidx_all
    uint32_t hot, x0;
};

// stable partition of every long row into its pieces (the entries keep their order inside the row)
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void bp_scatter_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                         const double *__restrict__ data,
                                                         const uint32_t *__restrict__ long_rows, uint64_t n_long,
                                                         const uint32_t *__restrict__ perm, SliceMap map, uint32_t npieces,
                                                         const uint64_t *__restrict__ pos,
                                                         const PieceBuild *__restrict__ pb,
                                                         double *__restrict__ vals_hot, uint16_t *__restrict__ cid_hot,
                                                         double *__restrict__ vals_cold, uint32_t *__restrict__ cid_cold) {
    __shared__ uint32_t fill[4][MAX_PIECES];       // entries of the row already placed, per piece
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    for (uint64_t j = w0; j < n_long; j += nw) {
        for (uint32_t k = lane; k < npieces; k += WAVE) fill[wave][k] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint64_t r = long_rows[j];
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        for (uint64_t p0 = s; p0 < e; p0 += WAVE) {
            const uint64_t p = p0 + lane;
            const bool valid = p < e;
            const uint32_t label = valid ? perm[indices[p]] : 0u;
            const double v = valid ? data[p] : 0.0;
            const uint32_t k = valid ? piece_of_label(map, label) : 0xFFFFFFFFu;
            // rank of the entry among the lanes of this batch that go to the same piece (lane order = row order)
            uint32_t rank = 0, group = 0;
            unsigned long long todo = __ballot(valid);
            while (todo) {                                  // wave-uniform: one round per distinct piece of the batch
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t kk = (uint32_t)__shfl((int)k, leader, WAVE);
                const unsigned long long m = __ballot(valid && k == kk);
                if (valid && k == kk) {
                    rank = (uint32_t)__popcll(m & below);
                    group = (uint32_t)__popcll(m);
                }
                todo &= ~m;
            }
            uint32_t before = 0;
            if (valid) before = fill[wave][k];
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) fill[wave][k] = before + group;   // one writer per piece
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const PieceBuild b = pb[k];
                const uint64_t e_rel = pos[(uint64_t)k * n_long + j] - b.start + before + rank;   // entry number inside the piece
                if (b.hot) {
                    const uint64_t tile = e_rel / WT;              // wave tile; entry i = 8 l + 2 p + e of it belongs to lane l
                    const uint32_t i = (uint32_t)(e_rel % WT);
                    const uint32_t ll = i / EPL, pp = (i % EPL) / 2, ee = i & 1u;
                    vals_hot[b.ent0 + tile * WT + pp * (WAVE * 2) + ll * 2 + ee] = v;
                    cid_hot[b.ent0 + e_rel] = (uint16_t)((label - b.x0) | (before + rank == 0 ? ROW_START : 0u));
                } else {
                    const uint64_t tile = e_rel / WT;
                    const uint32_t i = (uint32_t)(e_rel % WT);
                    const uint32_t ll = i / EPL, q = i % EPL;
                    vals_cold[b.ent0 + tile * WT + (q / 2) * (WAVE * 2) + ll * 2 + (q & 1u)] = v;
                    cid_cold[b.ent0 + tile * WT + (q / 4) * (WAVE * 4) + ll * 4 + (q & 3u)] =
                        label | (before + rank == 0 ? ROW_START32 : 0u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// compact row lists of the pieces: one thread per (piece, long row)
__global__ void bp_rows_kernel(const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ pos,
                               const uint64_t *__restrict__ pair, uint64_t n_long, uint32_t npieces,
                               const PieceBuild *__restrict__ pb, uint32_t *__restrict__ ptr_all,
                               uint32_t *__restrict__ rowidx_all) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)npieces * n_long) return;
    const uint32_t k = (uint32_t)(idx / n_long);
    const uint64_t j = idx - (uint64_t)k * n_long;
    const PieceBuild b = pb[k];
    if (j == 0) ptr_all[b.ptr_off + (pair[idx + n_long] - b.pair0)] = (uint32_t)b.nnz;    // end of the last row
    if (cnt[idx]) {
        const uint64_t q = pair[idx] - b.pair0;
        rowidx_all[b.row_off + q] = (uint32_t)j;
        ptr_all[b.ptr_off + q] = (uint32_t)(pos[idx] - b.start);
    }
}

// presence mask and first partial of every (block of 64 long rows, piece)
__global__ __launch_bounds__(256) void bp_wave_tables_kernel(const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ pair,
                                                             uint64_t n_long, uint32_t npieces, uint32_t np_pad, uint64_t nwb,
                                                             const PieceBuild *__restrict__ pb,
                                                             unsigned long long *__restrict__ wmask, uint32_t *__restrict__ wbase) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t wid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;     // = wb * npieces + k
    if (wid >= nwb * npieces) return;                                                  // wave-uniform
    const uint64_t wb = wid / npieces;
    const uint32_t k = (uint32_t)(wid - wb * npieces);
    const uint64_t j = wb * WAVE + lane;
    const bool have = j < n_long && cnt[(uint64_t)k * n_long + j] != 0;
    const unsigned long long m = __ballot(have);
    if (lane == 0) {
        const uint64_t slot = wb * np_pad + k;
        wmask[slot] = m;
        wbase[slot] = pb[k].row_off + (uint32_t)(pair[(uint64_t)k * n_long + wb * WAVE] - pb[k].pair0);
    }
}

struct TileRowJob {
    const uint32_t *ptr;
    uint32_t *tile_row;
    uint32_t nr, ntiles, T;
};

__global__ void bp_tile_rows_kernel(const TileRowJob *__restrict__ jobs) {
    const TileRowJob jb = jobs[blockIdx.y];
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > jb.ntiles) return;
    if (c == jb.ntiles) {
        jb.tile_row[c] = jb.nr;
        return;
    }
    const uint64_t target = (uint64_t)c * jb.T;
    uint32_t lo = 0, hi = jb.nr;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)jb.ptr[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    jb.tile_row[c] = lo;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct BandScratch {        // per stream
    double *partial = nullptr, *carry = nullptr, *xp = nullptr;
    BandPiece *pieces = nullptr;
    // the gather-bound kernels (cold pieces, short rows: L2 -> L1 fills) run beside the HBM-bound hot kernel on a second stream
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t cold_done = nullptr;                // the cold pieces' partial sums are written (the short rows may still run)
};

struct BandPlan {
    uint32_t nh = 0, phases = 1, npieces = 0;      // npieces = nh + 8 phases (the short piece comes after them)
    uint32_t xt_log2 = 14;                         // labels per hot slice = doubles of the x tile in LDS
    uint32_t n_long = 0, n_short_rows = 0;
    uint64_t cols = 0, cols_pad = 0;
    uint32_t *perm = nullptr, *long_rows = nullptr;
    uint32_t *inv_hot = nullptr;                   // inverse of the labelling for the labels of REFERENCED columns: inv_hot[l] = column
    uint32_t nref = 0;                             // referenced columns = labels in use ([nref, cols) are the columns nobody references)
    uint32_t hot_labels = 0;                       // nh << xt_log2, at most cols_pad
    double *vals_hot = nullptr, *vals_cold = nullptr;
    uint16_t *cid_hot = nullptr;
    uint32_t *cid_cold = nullptr;
    uint32_t *rowidx_all = nullptr, *tile_row_all = nullptr;
    Seg *segs = nullptr;                           // hot segments (workgroup by workgroup), then one segment per cold piece, the short piece
    HotSeg *hsegs = nullptr;                       // the hot segments again, as the hot kernel reads them (one record each)
    HotSeg *wg_first = nullptr;                    // per hot workgroup: its first segment's record with the segment range in the pad words
    uint32_t *wg_seg = nullptr;                    // hot workgroup b takes segments wg_seg[b] .. wg_seg[b + 1] - 1
    uint32_t nranges = 0, nsegs = 0, hot_wgs = 0, cold_tiles = 4, hot_run = 4;
    void *spills_y = nullptr;                      // Spill records (device) of the short rows: into y, by band_carry_kernel
    void *rspills = nullptr;                       // RSpill records (device) of the long rows, sorted by row: read by the reduction
    uint32_t *rsp_off = nullptr;                   // per block of 64 long rows: its first record (+ the end)
    uint32_t nspills = 0, nspills_y = 0;
    bool small = false;                            // few tiles per CU: the launches of one SpMV stay on one stream (the fork / join costs more than it hides)
    ColdGroup *groups = nullptr;
    unsigned long long *wmask = nullptr;           // per (64 long rows, piece): which rows have a partial
    uint32_t *wbase = nullptr;                     //                            and where the first one is
    uint32_t np_pad = 0;                           // table row length (npieces rounded up)
    uint64_t total_pairs = 0;
    uint32_t ngroups = 0, cold_blocks = 0, short_first_block = 0;
    bool has_short_group = false;
    std::vector<BandPiece> host_pieces;            // out filled per scratch
    std::vector<uint64_t> pair_off;
    std::unordered_map<void *, BandScratch> scratch;
    uint64_t bytes = 0;                            // HBM held by the plan (without scratch)
};

void band_free(BandPlan *bp) {
    if (!bp) return;
    auto drop = [](void *p) {
        if (p) (void)hipFree(p);
    };
    drop(bp->perm);
    drop(bp->inv_hot);
    drop(bp->long_rows);
    drop(bp->vals_hot);
    drop(bp->vals_cold);
    drop(bp->cid_hot);
    drop(bp->cid_cold);
    drop(bp->rowidx_all);
    drop(bp->tile_row_all);
    drop(bp->segs);
    drop(bp->hsegs);
    drop(bp->wg_first);
    drop(bp->rspills);
    drop(bp->rsp_off);
    drop(bp->spills_y);
    drop(bp->wg_seg);
    drop(bp->groups);
    drop(bp->wmask);
    drop(bp->wbase);
    for (auto &kv : bp->scratch) {
        drop(kv.second.partial);
        drop(kv.second.carry);
        drop(kv.second.xp);
        drop(kv.second.pieces);
        if (kv.second.cold_done) (void)hipEventDestroy(kv.second.cold_done);
        if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
        if (kv.second.join) (void)hipEventDestroy(kv.second.join);
        if (kv.second.aux) (void)hipStreamDestroy(kv.second.aux);
    }
    delete bp;
}

namespace {

struct PlanGuard {
    BandPlan *p;
    ~PlanGuard() { band_free(p); }
};

template <typename IDX, typename PTR>
int32_t band_build_t(sprs_hip_csmat *a, hipStream_t stream, BandPlan **out) {
    const Options &o = options();
    const uint64_t rows = a->rows, cols = a->cols, nnz = a->nnz;
    const PTR *ip = (const PTR *)a->indptr;
    const IDX *ix = (const IDX *)a->indices;
    *out = nullptr;
    if (rows >= 0xFFFFFFFFull || cols >= 0x7FFFFFFFull || !nnz) return SPRS_HIP_OK;   // bit 31 of a label flags a row start
    // rows with at least this many entries are "long" (cut into pieces)
    // (24 measured best on R-MAT 10M; a small matrix — fewer than ~400 hot tiles per CU, the `small` plans below — does better
    // with 8: R-MAT 1M 0.087 against 0.092 ms with two rounds of hot workgroups, profiles/r10u, r10v)
    const uint64_t split = o.spmv_band_split > 0 ? (uint64_t)o.spmv_band_split : (nnz < 400ull * 256ull * 512ull ? 8ull : 24ull);
    const uint32_t xt_log2 = o.spmv_band_tile == 8192 ? 13u : 14u;
    const uint64_t XT = 1ull << xt_log2;

    // ---- row classes --------------------------------------------------------------------
    TmpBuf short_flag, short_len, long_flag, short_pos, short_ptr, long_pos;
    SPRS_TRY_HIP(short_flag.alloc(rows * 8));
    SPRS_TRY_HIP(short_len.alloc(rows * 8));
    SPRS_TRY_HIP(long_flag.alloc(rows * 8));
    SPRS_TRY_HIP(short_pos.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(short_ptr.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(long_pos.alloc((rows + 1) * 8));
    hipLaunchKernelGGL(bp_classify_kernel<PTR>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, ip, rows, split,
                       short_flag.u64(), short_len.u64(), long_flag.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(short_flag.u64(), short_pos.u64(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(short_len.u64(), short_ptr.u64(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(long_flag.u64(), long_pos.u64(), rows, stream));
    uint64_t n_short_rows = 0, nnz_short = 0, n_long = 0;
    SPRS_TRY_HIP(hipMemcpy(&n_short_rows, short_pos.u64() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&nnz_short, short_ptr.u64() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&n_long, long_pos.u64() + rows, 8, hipMemcpyDeviceToHost));
    if (!n_long) return SPRS_HIP_OK;
    // auto mode: only when the long rows carry most of the entries (as the XCD-sliced plan)
    if (o.spmv_band == 0 && (nnz - nnz_short) * 2 < nnz) return SPRS_HIP_OK;
    if (nnz - nnz_short >= 0xFFFFFFFFull || nnz_short >= 0xFFFFFFFFull) return SPRS_HIP_OK;   // 32-bit piece offsets

    BandPlan *bp = new BandPlan();
    PlanGuard guard{bp};
    bp->cols = cols;
    bp->xt_log2 = xt_log2;
    bp->cols_pad = (cols + XT - 1) / XT * XT + XT;
    bp->n_long = (uint32_t)n_long;
    bp->n_short_rows = (uint32_t)n_short_rows;
    bp->cold_tiles = (uint32_t)(o.spmv_band_cold_tiles > 0 ? o.spmv_band_cold_tiles : 4);
    bp->hot_run = (uint32_t)(o.spmv_band_hot_run > 0 ? o.spmv_band_hot_run : 4);
    // 128 slices of 16384 labels measured best on R-MAT 10M (profiles/r05*: sweep over 64 .. 256)
    uint64_t nh = o.spmv_band_hot > 0 ? (uint64_t)o.spmv_band_hot : 128;
    if (nh > (cols + XT - 1) / XT) nh = (cols + XT - 1) / XT;
    if (nh > (uint64_t)MAX_HOT) nh = MAX_HOT;
    uint64_t phases = o.spmv_band_phases > 0 ? (uint64_t)o.spmv_band_phases : 1;
    if (phases > (uint64_t)MAX_PHASES) phases = MAX_PHASES;
    bp->nh = (uint32_t)nh;
    bp->phases = (uint32_t)phases;
    const uint32_t NP = (uint32_t)(nh + 8 * phases);
    bp->npieces = NP;
    SliceMap map;
    map.nh = (uint32_t)nh;
    map.phases = (uint32_t)phases;
    map.xt_log2 = xt_log2;
    map.hot_labels = nh << xt_log2;
    const uint64_t cold_labels = cols > map.hot_labels ? cols - map.hot_labels : 0;
    map.phase_width = (cold_labels + phases - 1) / phases;
    if (!map.phase_width) map.phase_width = 1;

    // the build needs four arrays of one word per (piece, long row): in auto mode a matrix whose long rows are too many for
    // that keeps the plans it had before this one existed
    const uint64_t flat = (uint64_t)NP * n_long;
    {
        size_t free_b = 0, total_b = 0;
        SPRS_TRY_HIP(hipMemGetInfo(&free_b, &total_b));
        if (o.spmv_band == 0 && flat * 32 + nnz * 12 > free_b / 2) return SPRS_HIP_OK;
    }

    // ---- labels ---------------------------------------------------------------------------
    uint64_t nref = 0;
    SPRS_TRY(build_column_labels<IDX>(ix, nnz, cols, stream, &bp->perm, &nref));
    bp->nref = (uint32_t)nref;
    bp->hot_labels = (uint32_t)(map.hot_labels < bp->cols_pad ? map.hot_labels : bp->cols_pad);
    SPRS_TRY_HIP(hipMalloc((void **)&bp->inv_hot, (nref + 4) * 4));
    SPRS_TRY_HIP(hipMemsetAsync(bp->inv_hot, 0, (nref + 4) * 4, stream));
    hipLaunchKernelGGL(bp_inverse_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, (const uint32_t *)bp->perm,
                       cols, bp->nref, bp->inv_hot);
    SPRS_TRY_HIP(hipGetLastError());

    // ---- short piece + list of long rows ------------------------------------------------------
    // cold arrays: [short piece | cold pieces], every piece starting at a multiple of 512 entries
    SPRS_TRY_HIP(hipMalloc((void **)&bp->long_rows, n_long * 4));

    // ---- long rows: count per piece, scans, placement --------------------------------------------
    TmpBuf cnt, nz, pos, pair, starts_d, pb_d;
    SPRS_TRY_HIP(cnt.alloc(flat * 8));
    SPRS_TRY_HIP(nz.alloc(flat * 8));
    SPRS_TRY_HIP(pos.alloc((flat + 1) * 8));
    SPRS_TRY_HIP(pair.alloc((flat + 1) * 8));
    // long_rows is needed by the count kernel: fill it (and the short piece's row lists) first; the short piece's entries
    // follow once the cold arrays exist
    TmpBuf s_rowidx_t, s_ptr_t;
    SPRS_TRY_HIP(s_rowidx_t.alloc((n_short_rows + 1) * 4));
    SPRS_TRY_HIP(s_ptr_t.alloc((n_short_rows + 1) * 4));
    hipLaunchKernelGGL((bp_fill_short_kernel<IDX, PTR>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream, ip, ix,
                       a->data, rows, short_pos.u64(), short_ptr.u64(), long_pos.u64(), bp->perm, (uint32_t *)s_rowidx_t.p,
                       (uint32_t *)s_ptr_t.p, (uint32_t *)nullptr, (double *)nullptr, bp->long_rows, split, 0, ShortBuckets{nullptr, 1, 1}, 0ull, n_long);
    SPRS_TRY_HIP(hipGetLastError());
    uint64_t wblocks = (n_long + 3) / 4;
    if (wblocks > 256 * 64) wblocks = 256 * 64;
    hipLaunchKernelGGL((bp_count_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(256), 0, stream, ip, ix, bp->long_rows,
                       n_long, bp->perm, map, NP, cnt.u64(), nz.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(cnt.u64(), pos.u64(), flat, stream));
    SPRS_TRY(exclusive_scan_u64(nz.u64(), pair.u64(), flat, stream));
    SPRS_TRY_HIP(starts_d.alloc(2 * (NP + 1) * 8));
    hipLaunchKernelGGL(bp_starts_kernel, dim3((NP + 256) / 256), dim3(256), 0, stream, pos.u64(), pair.u64(), n_long, NP,
                       starts_d.u64());
    SPRS_TRY_HIP(hipGetLastError());
    std::vector<uint64_t> starts(2 * (NP + 1));
    SPRS_TRY_HIP(hipMemcpy(starts.data(), starts_d.p, starts.size() * 8, hipMemcpyDeviceToHost));

    std::vector<PieceBuild> pb(NP);
    bp->host_pieces.assign(NP + 1, BandPiece());
    bp->pair_off.assign(NP + 1, 0);
    std::vector<uint64_t> tile_row_off(NP + 2, 0);
    // ---- the short piece's layout (it comes first in the cold arrays): ranges, buckets, padded size ------------------------
    int ncu_dev = 0;
    SPRS_TRY_HIP(hipDeviceGetAttribute(&ncu_dev, hipDeviceAttributeMultiprocessorCount, a->device));
    if (ncu_dev < 1) ncu_dev = 1;
    {   // cold pieces + short rows: few tiles (R-MAT 1M: 4 600) want one tile per wave, or the launch is a handful of waves per CU
        uint64_t gather_tiles = (nnz_short + WT - 1) / WT;
        for (uint32_t k = (uint32_t)nh; k < NP; ++k) gather_tiles += (starts[k + 1] - starts[k] + WT - 1) / WT;
        if (o.spmv_band_cold_tiles <= 0 && gather_tiles < 128ull * (uint64_t)ncu_dev) bp->cold_tiles = 1;
    }
    TmpBuf bucket0_d;
    ShortBuckets sbk{nullptr, 1, 1};
    uint64_t nnz_short_padded = nnz_short;
    {
        const uint64_t R = (uint64_t)bp->cold_tiles * WT;
        if (nnz_short && split >= 2 && (split - 1) * 16 <= R) {            // the gaps cost (split - 1) / R of the piece: at most 6 %
            const uint64_t Rp = R - (split - 1), nb = (nnz_short + Rp - 1) / Rp;
            SPRS_TRY_HIP(bucket0_d.alloc((nb + 1) * 8));
            hipLaunchKernelGGL(bp_bucket_starts_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, (const uint64_t *)short_ptr.u64(),
                               rows, Rp, nb, bucket0_d.u64());
            SPRS_TRY_HIP(hipGetLastError());
            uint64_t last0 = 0;
            SPRS_TRY_HIP(hipMemcpy(&last0, bucket0_d.u64() + (nb - 1), 8, hipMemcpyDeviceToHost));
            sbk = ShortBuckets{(const uint64_t *)bucket0_d.u64(), Rp, R};
            nnz_short_padded = (nb - 1) * R + (nnz_short - last0);          // (the last bucket may be empty: its first row is then "the end")
            if (nnz_short_padded >= 0xFFFFFFFFull) return SPRS_HIP_OK;
        }
    }
    uint64_t hot_tiles = 0, cold_ent = (nnz_short_padded + WT - 1) / WT * WT, ptr_off = 0, row_off = 0, tile_off = 0;   // the short piece comes first
    uint32_t max_tiles = 0;
    for (uint32_t k = 0; k < NP; ++k) {
        PieceBuild &b = pb[k];
        b.start = starts[k];
        b.nnz = starts[k + 1] - starts[k];
        b.pair0 = starts[NP + 1 + k];
        const uint64_t nr = starts[NP + 1 + k + 1] - b.pair0;
        b.hot = k < nh ? 1u : 0u;
        b.x0 = k < nh ? (uint32_t)((uint64_t)k << xt_log2) : 0u;
        b.ptr_off = (uint32_t)ptr_off;
        b.row_off = (uint32_t)row_off;
        bp->pair_off[k] = row_off;
        BandPiece &d = bp->host_pieces[k];
        d.nnz = b.nnz;
        d.nr = (uint32_t)nr;
        d.x0 = b.x0;
        d.to_y = 0;
        d.ntiles = (uint32_t)((b.nnz + WT - 1) / WT);             // wave tiles; every piece is padded to whole tiles
        if (b.hot) {
            b.ent0 = hot_tiles * WT;
            hot_tiles += d.ntiles;
        } else {
            b.ent0 = cold_ent;
            cold_ent += (uint64_t)d.ntiles * WT;
        }
        d.ent0 = b.ent0;
        tile_row_off[k] = tile_off;
        ptr_off += nr + 1;
        row_off += nr;
        tile_off += d.ntiles + 1;
        if (d.ntiles > max_tiles) max_tiles = d.ntiles;
    }
    {   // the short piece: index NP
        BandPiece &d = bp->host_pieces[NP];
        d.nnz = nnz_short_padded;
        d.nr = (uint32_t)n_short_rows;
        d.ntiles = (uint32_t)((nnz_short_padded + WT - 1) / WT);
        d.ent0 = 0;
        d.x0 = 0;
        d.to_y = 1;
        tile_row_off[NP] = tile_off;
        tile_off += d.ntiles + 1;
        tile_row_off[NP + 1] = tile_off;
        if (d.ntiles > max_tiles) max_tiles = d.ntiles;
    }
    if (ptr_off + n_short_rows + 1 >= 0xFFFFFFFFull || tile_off >= 0x7FFFFFFFull) return SPRS_HIP_OK;

    // ---- arrays of the plan ----------------------------------------------------------------------
    const uint64_t hot_entries = hot_tiles * WT;
    SPRS_TRY_HIP(hipMalloc((void **)&bp->vals_hot, (hot_entries + WT) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->cid_hot, (hot_entries + WT) * 2));
    SPRS_TRY_HIP(hipMemsetAsync(bp->vals_hot, 0, (hot_entries + WT) * 8, stream));
    SPRS_TRY_HIP(hipMemsetAsync(bp->cid_hot, 0, (hot_entries + WT) * 2, stream));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->vals_cold, (cold_ent + WT) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->cid_cold, (cold_ent + WT) * 4));
    // the padding of every piece — behind its last entry, and the gaps of the short piece — reads as (label `cols`, value 0):
    // xp[cols] is never written and stays 0.0, so a padding product is 0 * 0 whatever x holds
    SPRS_TRY_HIP(hipMemsetAsync(bp->vals_cold, 0, (cold_ent + WT) * 8, stream));
    SPRS_TRY_HIP(hipMemsetD32Async((hipDeviceptr_t)bp->cid_cold, (int)(uint32_t)cols, cold_ent + WT, stream));
    TmpBuf ptr_all;                                                // entry offsets of the compact rows: only the build reads them
    SPRS_TRY_HIP(ptr_all.alloc((ptr_off + n_short_rows + 2) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->rowidx_all, (row_off + n_short_rows + 1) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->tile_row_all, (tile_off + 1) * 4));
    bp->bytes = (hot_entries + WT) * 10 + (cold_ent + WT) * 12 + (row_off + n_short_rows + tile_off) * 4 + cols * 4 + n_long * 4 +
                ((uint64_t)bp->nref + 4) * 4;
    // short piece: its entries go straight into place (piece 0 of the cold arrays), its row lists are copied
    hipLaunchKernelGGL((bp_fill_short_kernel<IDX, PTR>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream, ip, ix,
                       a->data, rows, short_pos.u64(), short_ptr.u64(), long_pos.u64(), bp->perm, (uint32_t *)nullptr,
                       (uint32_t *)s_ptr_t.p, bp->cid_cold, bp->vals_cold, bp->long_rows, split, 1, sbk, nnz_short_padded, n_long);
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY_HIP(hipMemcpyAsync((uint32_t *)ptr_all.p + ptr_off, s_ptr_t.p, (n_short_rows + 1) * 4, hipMemcpyDeviceToDevice, stream));
    if (n_short_rows)
        SPRS_TRY_HIP(hipMemcpyAsync(bp->rowidx_all + row_off, s_rowidx_t.p, n_short_rows * 4, hipMemcpyDeviceToDevice, stream));

    SPRS_TRY_HIP(pb_d.alloc(NP * sizeof(PieceBuild)));
    SPRS_TRY_HIP(hipMemcpyAsync(pb_d.p, pb.data(), NP * sizeof(PieceBuild), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL((bp_scatter_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(256), 0, stream, ip, ix, a->data,
                       bp->long_rows, n_long, bp->perm, map, NP, pos.u64(), (const PieceBuild *)pb_d.p, bp->vals_hot,
                       bp->cid_hot, bp->vals_cold, bp->cid_cold);
    SPRS_TRY_HIP(hipGetLastError());
    hipLaunchKernelGGL(bp_rows_kernel, dim3((unsigned)((flat + 255) / 256)), dim3(256), 0, stream, cnt.u64(), pos.u64(),
                       pair.u64(), n_long, NP, (const PieceBuild *)pb_d.p, (uint32_t *)ptr_all.p, bp->rowidx_all);
    SPRS_TRY_HIP(hipGetLastError());

    {   // ---- tables of the final reduction ---------------------------------------------------------------
        const uint64_t nwb = (n_long + WAVE - 1) / WAVE;
        bp->total_pairs = row_off;
        bp->np_pad = (NP + (uint32_t)RU - 1u) / (uint32_t)RU * (uint32_t)RU;
        const uint64_t slots = nwb * bp->np_pad;
        SPRS_TRY_HIP(hipMalloc((void **)&bp->wmask, (slots + WAVE) * 8));
        SPRS_TRY_HIP(hipMalloc((void **)&bp->wbase, (slots + WAVE) * 4));
        SPRS_TRY_HIP(hipMemsetAsync(bp->wmask, 0, (slots + WAVE) * 8, stream));
        SPRS_TRY_HIP(hipMemsetAsync(bp->wbase, 0, (slots + WAVE) * 4, stream));
        const uint64_t waves = nwb * NP;
        hipLaunchKernelGGL(bp_wave_tables_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, cnt.u64(), pair.u64(),
                           n_long, NP, bp->np_pad, nwb, (const PieceBuild *)pb_d.p, bp->wmask, bp->wbase);
        SPRS_TRY_HIP(hipGetLastError());
        bp->bytes += slots * 12;
    }
    // ---- device pointers of the pieces, tile -> first row tables ------------------------------------
    std::vector<TileRowJob> jobs(NP + 1);
    for (uint32_t k = 0; k <= NP; ++k) {
        BandPiece &d = bp->host_pieces[k];
        const uint64_t po = k < NP ? pb[k].ptr_off : ptr_off, ro = k < NP ? pb[k].row_off : row_off;
        d.rowidx = bp->rowidx_all + ro;
        d.tile_row = bp->tile_row_all + tile_row_off[k];
        jobs[k] = TileRowJob{(const uint32_t *)ptr_all.p + po, bp->tile_row_all + tile_row_off[k], d.nr, d.ntiles, (uint32_t)WT};
    }
    TmpBuf jobs_d;
    SPRS_TRY_HIP(jobs_d.alloc(jobs.size() * sizeof(TileRowJob)));
    SPRS_TRY_HIP(hipMemcpyAsync(jobs_d.p, jobs.data(), jobs.size() * sizeof(TileRowJob), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bp_tile_rows_kernel, dim3((max_tiles + 256) / 256, NP + 1), dim3(256), 0, stream,
                       (const TileRowJob *)jobs_d.p);
    SPRS_TRY_HIP(hipGetLastError());

    // ---- who walks what: hot segments (equal shares of the hot tiles per workgroup), cold segments, launch groups ------------
    std::vector<Seg> segs;
    std::vector<uint32_t> wg_seg(1, 0u);
    uint32_t nranges = 0;
    auto add_seg = [&](uint32_t piece, uint32_t tile0, uint32_t n, uint32_t run) {
        segs.push_back(Seg{piece, tile0, n, nranges, run, 0u, 0u, 0u});
        nranges += (n + run - 1) / run;
    };
    if (hot_tiles) {
        int ncu = 0;
        SPRS_TRY_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, a->device));
        if (ncu < 1) ncu = 1;
        // workgroups per CU, ranges: a big matrix (R-MAT 10M: 2 180 tiles per CU) takes 2 rounds of long shares in ranges of 4 tiles
        // (median 1.048 against 1.072 ms with 1 round, profiles/r05j); a small one (R-MAT 1M: 104 tiles per CU) is latency
        // bound — every wave gets only a few tiles — and does better with more, shorter shares and single-tile ranges
        // (hot kernel 45 against 61 us, profiles/r05p)
        const bool small_hot = hot_tiles < 400ull * (uint64_t)ncu;
        // (small plans: 4 rounds until round 4; 2 measured better with split 8, profiles/r10u.  Big plans: 2 until round 5; with the
        // hot kernel no longer waiting for its stores 3 rounds measured 3 - 5 % better on three boxes — 1.046 / 1.049 - 1.089 / 1.059 - 1.098
        // against 1.081 / 1.114 - 1.117 / 1.080 - 1.116 ms — and 4 rounds 5 % worse, profiles/r13b, r13n, r13o)
        // (round 6, small plans: with the shares cut by modelled cost ONE round is best — 0.0717 against 0.0731 ms with two and 0.0784 with
        // three on R-MAT 1M, profiles/r15d: every workgroup pays ~5 us for its first x tile, 256 CUs asking for theirs at once)
        const uint64_t rounds = o.spmv_band_rounds > 0 ? (uint64_t)o.spmv_band_rounds
                                : (small_hot ? (o.spmv_band_balance == 2 ? 2 : 1) : 3);
        // (small plans: single-tile ranges until round 6; with one balanced round a workgroup holds ~115 tiles, 7 per wave, and ranges of
        // two halve the heads the reduction reads: R-MAT 1M 0.0682 - 0.0687 against 0.0699 - 0.0701 ms, ranges of four 0.0768, gpurun_out/r15m)
        if (o.spmv_band_hot_run <= 0 && small_hot) bp->hot_run = 2;
        bp->small = small_hot;
        // SHARES BY COST (small plans; option spmv_band_balance): a tile of a late slice holds ten times the row ends of an early one
        // and takes up to three times as long (every row end is a partial sum to scan, stage and store: 658 us against 470 us without
        // any on R-MAT 10M's 72 per tile), so equal tile counts leave the workgroups of the late slices running when the others are
        // done.  Beside the gather kernels of a big plan that tail is filled (profiles/r06a: hot kernel alone 681 against 736 us, the
        // overlapped SpMV unchanged); a small plan's hot kernel runs ALONE and its tail is the SpMV's.  cost(tile) = 180 + row ends.
        std::vector<uint64_t> cost_pre;                               // cost of the hot tiles before tile t (slice after slice)
        const bool balance = o.spmv_band_balance == 1 || (o.spmv_band_balance == 0 && small_hot);
        if (balance && o.spmv_band_share <= 0) {
            SPRS_TRY_HIP(hipStreamSynchronize(stream));               // bp_tile_rows_kernel
            std::vector<uint32_t> tr(tile_off + 1);
            SPRS_TRY_HIP(hipMemcpy(tr.data(), bp->tile_row_all, tr.size() * 4, hipMemcpyDeviceToHost));
            cost_pre.assign(1, 0ull);
            for (uint32_t k = 0; k < nh; ++k) {
                const uint32_t *t = tr.data() + tile_row_off[k];
                for (uint32_t c = 0; c < bp->host_pieces[k].ntiles; ++c) cost_pre.push_back(cost_pre.back() + 180ull + (uint64_t)(t[c + 1] - t[c]));
            }
        }
        auto split = [&](uint32_t k_lo, uint32_t k_hi) {             // equal shares of the tiles of slices [k_lo, k_hi) per workgroup
            uint64_t tiles = 0;
            for (uint32_t k = k_lo; k < k_hi; ++k) tiles += bp->host_pieces[k].ntiles;
            if (!tiles) return;
            uint64_t nwg = (uint64_t)ncu * rounds;
            if (nwg > tiles) nwg = tiles;
            uint64_t Q = (tiles + nwg - 1) / nwg;                     // wave tiles per workgroup
            if (o.spmv_band_share > 0) Q = (uint64_t)o.spmv_band_share;   // (A/B: the share sets how far apart the workgroups stream)
            // share b = tiles [cut(b), cut(b + 1)): equal counts, or equal cost (the first tile whose cost prefix reaches b / nwg of the total)
            const uint64_t nshares = cost_pre.empty() ? (tiles + Q - 1) / Q : nwg;
            auto cut = [&](uint64_t b) -> uint64_t {
                if (cost_pre.empty()) return std::min(tiles, b * Q);
                if (b >= nshares) return tiles;
                const uint64_t target = (uint64_t)((long double)cost_pre.back() * (long double)b / (long double)nshares);
                return (uint64_t)(std::lower_bound(cost_pre.begin(), cost_pre.end(), target) - cost_pre.begin());
            };
            uint32_t k = k_lo;
            uint64_t k_first = 0;                                     // number (inside the group) of slice k's first tile
            for (uint64_t b = 0; b < nshares; ++b) {
                uint64_t t = cut(b);
                const uint64_t t_end = std::min(tiles, cut(b + 1));
                if (t >= t_end) continue;                             // (an empty share: more workgroups than tiles of that cost)
                while (t < t_end) {
                    while (k_first + bp->host_pieces[k].ntiles <= t) k_first += bp->host_pieces[k++].ntiles;   // (slices without tiles are skipped)
                    const uint64_t s_end = std::min(t_end, k_first + bp->host_pieces[k].ntiles);
                    add_seg(k, (uint32_t)(t - k_first), (uint32_t)(s_end - t), bp->hot_run);
                    t = s_end;
                }
                wg_seg.push_back((uint32_t)segs.size());
            }
        };
        split(0, (uint32_t)nh);
        bp->hot_wgs = (uint32_t)(wg_seg.size() - 1);
    }
    for (uint32_t k = (uint32_t)nh; k <= NP; ++k) {
        BandPiece &d = bp->host_pieces[k];
        d.range0 = nranges;
        if (d.ntiles) add_seg(k, 0u, d.ntiles, bp->cold_tiles);
    }
    bp->nranges = nranges;
    bp->nsegs = (uint32_t)segs.size();
    SPRS_TRY_HIP(hipMalloc((void **)&bp->segs, (segs.size() + 1) * sizeof(Seg)));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->wg_seg, wg_seg.size() * 4));
    if (!segs.empty()) SPRS_TRY_HIP(hipMemcpyAsync(bp->segs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice, stream));
    SPRS_TRY_HIP(hipMemcpyAsync(bp->wg_seg, wg_seg.data(), wg_seg.size() * 4, hipMemcpyHostToDevice, stream));
    bp->bytes += segs.size() * sizeof(Seg);
    if (bp->hot_wgs) {
        const uint32_t nhs = wg_seg.back();                              // the hot segments come first
        std::vector<HotSeg> hs(nhs);
        for (uint32_t i = 0; i < nhs; ++i) {
            const Seg &sg = segs[i];
            const BandPiece &d = bp->host_pieces[sg.piece];
            hs[i] = HotSeg{d.ent0, d.nnz, d.tile_row, bp->pair_off[sg.piece], d.x0, sg.tile0, sg.ntiles, sg.range0, sg.run, 0u, 0u, 0u};
        }
        std::vector<HotSeg> wf(bp->hot_wgs);
        for (uint32_t b = 0; b < bp->hot_wgs; ++b) {
            wf[b] = hs[wg_seg[b]];
            wf[b].pad0 = wg_seg[b];
            wf[b].pad1 = wg_seg[b + 1];
        }
        SPRS_TRY_HIP(hipMalloc((void **)&bp->wg_first, (wf.size() + 1) * sizeof(HotSeg)));
        SPRS_TRY_HIP(hipMemcpyAsync(bp->wg_first, wf.data(), wf.size() * sizeof(HotSeg), hipMemcpyHostToDevice, stream));
        SPRS_TRY_HIP(hipMalloc((void **)&bp->hsegs, (hs.size() + 1) * sizeof(HotSeg)));
        SPRS_TRY_HIP(hipMemcpyAsync(bp->hsegs, hs.data(), hs.size() * sizeof(HotSeg), hipMemcpyHostToDevice, stream));
        SPRS_TRY_HIP(hipStreamSynchronize(stream));                      // (hs goes out of scope)
        bp->bytes += (hs.size() + wf.size()) * sizeof(HotSeg);
    }

    std::vector<ColdGroup> groups;
    uint32_t blocks = 0;
    const uint32_t wpb = CNT / WAVE, per_block = wpb * bp->cold_tiles;      // tiles a cold workgroup walks
    for (uint32_t ph = 0; ph < phases; ++ph) {
        uint32_t mt = 0;
        for (uint32_t s = 0; s < 8; ++s) mt = std::max(mt, (bp->host_pieces[nh + ph * 8 + s].ntiles + per_block - 1) / per_block);
        if (!mt) continue;
        groups.push_back(ColdGroup{blocks, (uint32_t)(nh + ph * 8), 8});
        blocks += mt * 8;
    }
    if (bp->host_pieces[NP].ntiles) {
        bp->short_first_block = blocks;
        bp->has_short_group = true;
        groups.push_back(ColdGroup{blocks, NP, 1});
        blocks += (bp->host_pieces[NP].ntiles + per_block - 1) / per_block;
    }
    bp->ngroups = (uint32_t)groups.size();
    bp->cold_blocks = blocks;
    if (bp->ngroups) {
        SPRS_TRY_HIP(hipMalloc((void **)&bp->groups, groups.size() * sizeof(ColdGroup)));
        SPRS_TRY_HIP(hipMemcpyAsync(bp->groups, groups.data(), groups.size() * sizeof(ColdGroup), hipMemcpyHostToDevice, stream));
    }
    // ---- rows that run on from one range into the next: one record per run -----------------------------------------------
    if (nranges) {
        TmpBuf pcs_d, poff_d, cnt_d;
        SPRS_TRY_HIP(pcs_d.alloc(bp->host_pieces.size() * sizeof(BandPiece)));
        SPRS_TRY_HIP(poff_d.alloc(bp->pair_off.size() * 8));
        SPRS_TRY_HIP(cnt_d.alloc(8));
        SPRS_TRY_HIP(hipMemcpyAsync(pcs_d.p, bp->host_pieces.data(), bp->host_pieces.size() * sizeof(BandPiece), hipMemcpyHostToDevice, stream));
        SPRS_TRY_HIP(hipMemcpyAsync(poff_d.p, bp->pair_off.data(), bp->pair_off.size() * 8, hipMemcpyHostToDevice, stream));
        SPRS_TRY_HIP(hipMemsetAsync(cnt_d.p, 0, 8, stream));
        TmpBuf sp_tmp;                                                                       // the long rows' records, as found (any order)
        SPRS_TRY_HIP(sp_tmp.alloc(((uint64_t)nranges + 1) * sizeof(Spill)));                 // at most one per range
        SPRS_TRY_HIP(hipMalloc(&bp->spills_y, ((uint64_t)nranges + 1) * sizeof(Spill)));
        hipLaunchKernelGGL(bp_spill_kernel, dim3((nranges + 255) / 256), dim3(256), 0, stream, (const Seg *)bp->segs, bp->nsegs, nranges,
                           (const BandPiece *)pcs_d.p, (const uint64_t *)poff_d.p, bp->nh, (const uint16_t *)bp->cid_hot,
                           (const uint32_t *)bp->cid_cold, (Spill *)bp->spills_y, (Spill *)sp_tmp.p,
                           (unsigned int *)cnt_d.p);
        SPRS_TRY_HIP(hipGetLastError());
        uint32_t counts[2] = {0, 0};
        SPRS_TRY_HIP(hipMemcpy(counts, cnt_d.p, 8, hipMemcpyDeviceToHost));
        bp->nspills_y = counts[0];
        bp->nspills = counts[1];
        if (bp->nspills) {
            // the reduction reads them by row block: sorted by (long row, first carry slot) — the slot order is the order of the
            // ranges, so a row's heads are added piece by piece, tile by tile — and indexed per block of 64 long rows
            std::vector<Spill> found(bp->nspills);
            SPRS_TRY_HIP(hipMemcpy(found.data(), sp_tmp.p, found.size() * sizeof(Spill), hipMemcpyDeviceToHost));
            std::sort(found.begin(), found.end(), [](const Spill &a, const Spill &b) { return a.j != b.j ? a.j < b.j : a.first < b.first; });
            const uint32_t nwb = (uint32_t)((n_long + WAVE - 1) / WAVE);
            // (a lane adds the carries of its record one after the other: a hub row's run of 40 ranges is cut into records of 8)
            std::vector<RSpill> recs;
            recs.reserve(found.size() + found.size() / 8);
            std::vector<uint32_t> off(nwb + 1, 0u);
            for (size_t i = 0; i < found.size(); ++i) {
                for (uint32_t k = 0; k < found[i].n; k += 8) {
                    recs.push_back(RSpill{found[i].j, found[i].first + k, std::min(8u, found[i].n - k), 0u});
                    ++off[found[i].j / WAVE + 1];
                }
            }
            for (uint32_t w = 0; w < nwb; ++w) off[w + 1] += off[w];
            SPRS_TRY_HIP(hipMalloc(&bp->rspills, recs.size() * sizeof(RSpill)));
            SPRS_TRY_HIP(hipMalloc((void **)&bp->rsp_off, off.size() * 4));
            SPRS_TRY_HIP(hipMemcpy(bp->rspills, recs.data(), recs.size() * sizeof(RSpill), hipMemcpyHostToDevice));
            SPRS_TRY_HIP(hipMemcpy(bp->rsp_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
            bp->bytes += recs.size() * sizeof(RSpill) + off.size() * 4;
        }
        bp->bytes += (uint64_t)counts[0] * sizeof(Spill);
    }
    SPRS_TRY_HIP(hipStreamSynchronize(stream));   // plan complete, temporaries (and the host vectors above) may go
    if (getenv("SPRS_HIP_DEBUG")) {
        uint64_t hot_nnz = 0, cold_nnz = 0, hot_pairs = 0, cold_pairs = 0;
        for (uint32_t k = 0; k < NP; ++k) {
            (k < nh ? hot_nnz : cold_nnz) += bp->host_pieces[k].nnz;
            (k < nh ? hot_pairs : cold_pairs) += bp->host_pieces[k].nr;
        }
        fprintf(stderr, "[sprs_hip band] rows %llu nnz %llu | long rows %llu, short non-empty rows %llu with %llu entries | hot: %u slices of %llu labels, "
                        "%llu entries, %llu (row,slice) pairs, %u workgroups, %u segments | cold: %u pieces, %llu entries, %llu pairs | %u ranges, %u + %u heads (long rows + short rows), short piece %llu entries laid out in %llu | plan %.1f MB\n",
                (unsigned long long)rows, (unsigned long long)nnz, (unsigned long long)n_long, (unsigned long long)n_short_rows,
                (unsigned long long)nnz_short, (unsigned)nh, (unsigned long long)XT, (unsigned long long)hot_nnz, (unsigned long long)hot_pairs,
                bp->hot_wgs, bp->nsegs, (unsigned)(8 * phases), (unsigned long long)cold_nnz, (unsigned long long)cold_pairs, bp->nranges, bp->nspills, bp->nspills_y,
                (unsigned long long)nnz_short, (unsigned long long)nnz_short_padded, bp->bytes / 1e6);
    }
    guard.p = nullptr;
    *out = bp;
    return SPRS_HIP_OK;
}

int32_t band_scratch(BandPlan *bp, hipStream_t stream, BandScratch **out) {
    auto it = bp->scratch.find((void *)stream);
    if (it == bp->scratch.end()) {
        BandScratch sc;
        SPRS_TRY_HIP(hipMalloc((void **)&sc.partial, (bp->total_pairs + 1) * 8));   // every pair is written by every SpMV
        SPRS_TRY_HIP(hipMalloc((void **)&sc.carry, ((uint64_t)bp->nranges + 1) * 8));
        SPRS_TRY_HIP(hipMalloc((void **)&sc.xp, bp->cols_pad * 8));
        SPRS_TRY_HIP(hipMemset(sc.xp, 0, bp->cols_pad * 8));      // the padding behind the last column is read into LDS
        std::vector<BandPiece> pcs = bp->host_pieces;
        for (uint32_t k = 0; k <= bp->npieces; ++k) pcs[k].out = k < bp->npieces ? sc.partial + bp->pair_off[k] : nullptr;
        SPRS_TRY_HIP(hipMalloc((void **)&sc.pieces, pcs.size() * sizeof(BandPiece)));
        SPRS_TRY_HIP(hipMemcpy(sc.pieces, pcs.data(), pcs.size() * sizeof(BandPiece), hipMemcpyHostToDevice));
        SPRS_TRY_HIP(hipStreamCreateWithFlags(&sc.aux, hipStreamNonBlocking));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&sc.fork, hipEventDisableTiming));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&sc.join, hipEventDisableTiming));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&sc.cold_done, hipEventDisableTiming));
        it = bp->scratch.emplace((void *)stream, sc).first;
    }
    *out = &it->second;
    return SPRS_HIP_OK;
}

}  // namespace

int32_t band_build(sprs_hip_csmat *a, hipStream_t stream, BandPlan **out) {
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return band_build_t<uint64_t, uint64_t>(a, stream, out);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return band_build_t<uint32_t, uint64_t>(a, stream, out);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return band_build_t<uint64_t, uint32_t>(a, stream, out);
    return band_build_t<uint32_t, uint32_t>(a, stream, out);
}

uint64_t band_plan_bytes(const BandPlan *bp) { return bp ? bp->bytes : 0; }

// One SpMV on a banded plan: the launches, on `stream` and on the scratch's second stream.
//   stream: gather the hot labels of x -> [fork] -> hot slices ............ [cold_done] -> carries -> reduce ........ [join]
//   aux:                                  [fork] -> scatter the rest of x, clear y -> cold pieces -> [cold_done] -> short rows -> their carries -> [join]
// The gather-bound launches (cold pieces + short rows: L2 -> L1 line fills) run beside the HBM-bound hot slices (option
// spmv_band_overlap, 2 = off); with the overlap the permutation is split as well (spmv_band_split_permute, 2 = off).
// The reduction of the long rows needs the hot slices and the cold pieces, NOT the short rows (they write y themselves): it
// starts when the hot kernel ends and runs beside what is left of the short rows (round 5; until then it waited for the whole
// second stream; option spmv_band_tail = 2 brings that order back for A/B).  What it is worth is small: concurrent kernels of
// this SpMV mostly time-share the fabric — the sum of their times alone is 1 190 us, side by side they take 1 040 - 1 080
// (profiles/r13c, r13d) — and holding part of the short rows back until the hot kernel is done, so that they run beside the
// reduction, was SLOWER (1.11 - 1.16 against 1.07 ms, profiles/r13d: a small kernel like the carries then waits behind them).
int32_t band_spmv(sprs_hip_csmat *a, BandPlan *bp, const double *x, double *y, bool acc, hipStream_t stream) {
    BandScratch *sc = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lock(a->mu);
        SPRS_TRY(band_scratch(bp, stream, &sc));
    }
    // (a small plan keeps to one stream unless the overlap is asked for: R-MAT 1M 0.092 against 0.096 ms, profiles/r05p)
    const bool overlap = options().spmv_band_overlap != 2 && bp->hot_wgs && bp->cold_blocks && (!bp->small || options().spmv_band_overlap == 1);
    const bool split_permute = overlap && options().spmv_band_split_permute != 2 && bp->hot_labels;
    const bool early_reduce = overlap && options().spmv_band_tail != 2;
    hipStream_t cstream = overlap ? sc->aux : stream;
    // the short rows' heads go into y behind them (the long rows' are read by the reduction itself)
    auto launch_carry_y = [&](hipStream_t st) -> int32_t {
        const uint32_t n = bp->nspills_y;
        if (!n) return SPRS_HIP_OK;
        hipLaunchKernelGGL(band_carry_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, st, (const Spill *)bp->spills_y, n, (const double *)sc->carry,
                           sc->partial, y);
        SPRS_TRY_HIP(hipGetLastError());
        return SPRS_HIP_OK;
    };
    const uint32_t cut = bp->has_short_group ? bp->short_first_block : bp->cold_blocks;
    // small plans on one stream: the short rows share the reduction's launch (band_tail_kernel); spmv_band_tail = 2: their own launch in
    // front of the hot slices (round 5, A/B)
    const bool fused_tail = !overlap && bp->small && options().spmv_band_tail != 2 && bp->cold_blocks > cut;
    auto launch_hot = [&]() -> int32_t {
        if (!bp->hot_wgs) return SPRS_HIP_OK;
        const uint32_t lds = hot_lds_bytes(bp->xt_log2);
#ifndef SPRS_HIP_EMU
        {   // more than 64 KiB of dynamic LDS has to be asked for, once per kernel and device
            static std::mutex mu;
            static std::unordered_map<int, bool> done;
            std::lock_guard<std::mutex> lock(mu);
            if (!done[a->device]) {
                SPRS_TRY_HIP(hipFuncSetAttribute((const void *)band_hot_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hot_lds_bytes(13)));
                SPRS_TRY_HIP(hipFuncSetAttribute((const void *)band_hot_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hot_lds_bytes(14)));
                done[a->device] = true;
            }
        }
#endif
        unsigned long long *prof = nullptr;
        if (DEVTOOLS && (options().spmv_band_debug & 16)) {
            static unsigned long long *buf = nullptr;
            static uint32_t cap = 0;
            if (cap < bp->hot_wgs) {
                if (buf) (void)hipFree(buf);
                cap = bp->hot_wgs;
                SPRS_TRY_HIP(hipMalloc((void **)&buf, (size_t)cap * 24));
            }
            prof = buf;
        }
        const uint32_t xcd_shares = options().spmv_band_xcd == 1 || (options().spmv_band_xcd == 0 && bp->small) ? 1u : 0u;
        if (bp->xt_log2 == 13)
            hipLaunchKernelGGL((band_hot_kernel<13>), dim3(bp->hot_wgs), dim3(HOT_THREADS), lds, stream, (const HotSeg *)bp->hsegs,
                               (const HotSeg *)bp->wg_first,
                               (const double *)bp->vals_hot, (const uint16_t *)bp->cid_hot, (const double *)sc->xp, sc->partial, sc->carry,
                               (uint32_t)options().spmv_band_debug, prof, xcd_shares);
        else
            hipLaunchKernelGGL((band_hot_kernel<14>), dim3(bp->hot_wgs), dim3(HOT_THREADS), lds, stream, (const HotSeg *)bp->hsegs,
                               (const HotSeg *)bp->wg_first,
                               (const double *)bp->vals_hot, (const uint16_t *)bp->cid_hot, (const double *)sc->xp, sc->partial, sc->carry,
                               (uint32_t)options().spmv_band_debug, prof, xcd_shares);
        SPRS_TRY_HIP(hipGetLastError());
        if (DEVTOOLS && prof && getenv("SPRS_HIP_HOTPROF")) {      // (synchronous: a developer printout, not a timing run)
            SPRS_TRY_HIP(hipStreamSynchronize(stream));
            std::vector<unsigned long long> t((size_t)bp->hot_wgs * 3);
            SPRS_TRY_HIP(hipMemcpy(t.data(), prof, t.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (uint32_t b = 0; b < bp->hot_wgs; ++b) {
                t0 = std::min(t0, t[3 * b]);
                t1 = std::max(t1, t[3 * b + 2]);
            }
            std::vector<double> st, xw, du, en;
            for (uint32_t b = 0; b < bp->hot_wgs; ++b) {
                st.push_back((t[3 * b] - t0) * 0.01);
                xw.push_back((t[3 * b + 1] - t[3 * b]) * 0.01);
                du.push_back((t[3 * b + 2] - t[3 * b]) * 0.01);
                en.push_back((t[3 * b + 2] - t0) * 0.01);
            }
            auto q = [](std::vector<double> v, double f) {
                std::sort(v.begin(), v.end());
                return v[(size_t)(f * (v.size() - 1))];
            };
            fprintf(stderr, "[hot_prof] %u workgroups, span %.1f us | start min/med/p90/max %.1f %.1f %.1f %.1f | first x tile after %.1f %.1f %.1f %.1f | "
                            "duration %.1f %.1f %.1f %.1f | end %.1f %.1f %.1f %.1f\n", bp->hot_wgs, (t1 - t0) * 0.01,
                    q(st, 0), q(st, .5), q(st, .9), q(st, 1), q(xw, 0), q(xw, .5), q(xw, .9), q(xw, 1), q(du, 0), q(du, .5), q(du, .9), q(du, 1),
                    q(en, 0), q(en, .5), q(en, .9), q(en, 1));
        }
        return SPRS_HIP_OK;
    };
    // blocks [b0, b0 + nb) of the gather launch: cold pieces (partial sums out) below `cut`, short rows (y out) from there on
    auto cold_args = [&](uint32_t b0) {
        return ColdArgs{sc->pieces, bp->groups, bp->ngroups, bp->vals_cold, bp->cid_cold, sc->xp, y, sc->carry, b0, bp->cold_tiles, 0u, BandPiece()};
    };
    auto launch_gather = [&](uint32_t b0, uint32_t nb) -> int32_t {
        if (!nb) return SPRS_HIP_OK;
        const ColdArgs ca = cold_args(b0);
        if (b0 < cut) hipLaunchKernelGGL((band_cold_kernel<false, false>), dim3(nb), dim3(CNT), 0, cstream, ca);
        else if (acc) hipLaunchKernelGGL((band_cold_kernel<true, true>), dim3(nb), dim3(CNT), 0, cstream, ca);
        else hipLaunchKernelGGL((band_cold_kernel<false, true>), dim3(nb), dim3(CNT), 0, cstream, ca);
        SPRS_TRY_HIP(hipGetLastError());
        return SPRS_HIP_OK;
    };

    // x into the plan's labelling (referenced columns only) and y cleared: one launch; with the overlap the hot labels first, the
    // rest and y on the second stream beside the hot kernel (option spmv_band_split_permute)
    const bool by_gather = (uint64_t)bp->nref * 3 <= bp->cols;        // (see band_gather_kernel)
    auto launch_xp = [&](uint32_t l0, uint32_t l1, bool clear_y, hipStream_t st, bool hot_only = false) -> int32_t {
        if (!by_gather && !hot_only) {                                   // the scatter form; a big plan's hot labels are always gathered
            const uint64_t span = clear_y ? std::max<uint64_t>(bp->cols, a->rows) : bp->cols;
            hipLaunchKernelGGL(band_permute_kernel, dim3((unsigned)((span + 1023) / 1024)), dim3(256), 0, st, x, (const uint32_t *)bp->perm, bp->cols,
                               sc->xp, clear_y ? y : (double *)nullptr, a->rows, l0, (uint32_t)bp->cols_pad);     // (every label from l0 on: leaving out the
            // unreferenced columns' stores made R-MAT 10M 5 - 8 % SLOWER on average — 1.10 - 1.14 against 1.05 ms, same best step —, gpurun_out/r15x)
            SPRS_TRY_HIP(hipGetLastError());
            return SPRS_HIP_OK;
        }
        const uint64_t n4 = std::max<uint64_t>(l1 > l0 ? ((uint64_t)(l1 - l0) + 3) / 4 : 0, clear_y ? (a->rows + 3) / 4 : 0);
        if (!n4) return SPRS_HIP_OK;
        if (clear_y)
            hipLaunchKernelGGL(band_gather_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, (const uint32_t *)bp->inv_hot, l0,
                               l1 > l0 ? l1 : l0, sc->xp, y, a->rows);
        else
            hipLaunchKernelGGL(band_gather_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, (const uint32_t *)bp->inv_hot, l0,
                               l1 > l0 ? l1 : l0, sc->xp, (double *)nullptr, a->rows);
        SPRS_TRY_HIP(hipGetLastError());
        return SPRS_HIP_OK;
    };
    const uint32_t hot_end = bp->hot_labels < bp->nref ? bp->hot_labels : bp->nref;
    if (split_permute) SPRS_TRY(launch_xp(0u, hot_end, false, stream, true));
    else SPRS_TRY(launch_xp(0u, bp->nref, !acc, stream));
    if (overlap) {
        SPRS_TRY_HIP(hipEventRecord(sc->fork, stream));           // the hot labels of xp (or all of it, and the cleared y) are ready
        SPRS_TRY(launch_hot());
        SPRS_TRY_HIP(hipStreamWaitEvent(sc->aux, sc->fork, 0));
    }
    if (split_permute) SPRS_TRY(launch_xp(hot_end, bp->nref, !acc, cstream));
    SPRS_TRY(launch_gather(0, cut));
    if (early_reduce) SPRS_TRY_HIP(hipEventRecord(sc->cold_done, sc->aux));
    if (!fused_tail) SPRS_TRY(launch_gather(cut, bp->cold_blocks - cut));
    // the short rows' own carries follow them on their stream
    if (early_reduce) SPRS_TRY(launch_carry_y(sc->aux));
    if (overlap) SPRS_TRY_HIP(hipEventRecord(sc->join, sc->aux));
    else SPRS_TRY(launch_hot());
    if (overlap) SPRS_TRY_HIP(hipStreamWaitEvent(stream, early_reduce ? sc->cold_done : sc->join, 0));
    if (!early_reduce && !fused_tail) SPRS_TRY(launch_carry_y(stream));
    const uint32_t nwb = (bp->n_long + WAVE - 1) / WAVE;
    const uint32_t per_xcd = (nwb + 7) / 8;
    const dim3 rg(((per_xcd + 3) / 4) * 8), rb(256);                 // reduction: one wave per block of 64 long rows, XCD by XCD
    const ReduceArgs ra{sc->partial, bp->wmask, bp->wbase, bp->long_rows, (const RSpill *)bp->rspills, bp->nspills ? bp->rsp_off : nullptr,
                        sc->carry, y, bp->n_long, bp->np_pad, nwb};
    if (fused_tail) {
        ColdArgs ca = cold_args(cut);
        ca.direct = 1u;                                         // the short piece by value: the blocks from `cut` on walk only it
        ca.piece = bp->host_pieces[bp->npieces];
        const dim3 tg(rg.x + (bp->cold_blocks - cut));
        if (acc) hipLaunchKernelGGL(band_tail_kernel<true>, tg, rb, 0, stream, ra, rg.x, ca);
        else hipLaunchKernelGGL(band_tail_kernel<false>, tg, rb, 0, stream, ra, rg.x, ca);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY(launch_carry_y(stream));                       // (no records unless the short rows are long enough to forbid the gapped layout)
    } else {
        if (acc) hipLaunchKernelGGL(band_reduce_kernel<true>, rg, rb, 0, stream, ra);
        else hipLaunchKernelGGL(band_reduce_kernel<false>, rg, rb, 0, stream, ra);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (early_reduce) SPRS_TRY_HIP(hipStreamWaitEvent(stream, sc->join, 0));   // the short rows and their carries
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
