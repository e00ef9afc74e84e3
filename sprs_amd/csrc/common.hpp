// Internal definitions shared by the translation units of libsprs_hip.so.
// Public surface: include/sprs_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/sprs_hip.h"

struct sprs_hip_spgemm_plan;      // spgemm.hip
struct sprs_hip_dist;             // dist.hip

namespace sprs_hip {

// ---- thread-local error state ------------------------------------------
void set_error(int32_t status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int32_t fail_hip(hipError_t e, const char *what);   // records + returns status
void clear_error();

#define SPRS_TRY_HIP(expr)                                              \
    do {                                                                \
        hipError_t e__ = (expr);                                        \
        if (e__ != hipSuccess) return ::sprs_hip::fail_hip(e__, #expr); \
    } while (0)

#define SPRS_TRY(expr)                      \
    do {                                    \
        int32_t s__ = (expr);               \
        if (s__ != SPRS_HIP_OK) return s__; \
    } while (0)

#define SPRS_FAIL(status, ...)                        \
    do {                                              \
        ::sprs_hip::set_error((status), __VA_ARGS__); \
        return (status);                              \
    } while (0)

// ---- options -------------------------------------------------------------
struct Options {
    int64_t spmv_kernel = 0;       // 0 auto, 1 nnz-tiled, 2 wave-per-row (A/B only)
    int64_t spmv_xcs = 0;          // XCD-sliced plan for long rows: 0 auto, 1 force on, 2 off
    int64_t spmv_xcs_split = 32;   // rows with >= this many entries go to the sliced part
    int64_t spmv_xcs_idx32 = 1;    // plan-owned copies store 32-bit column ids when cols < 2^32
    int64_t spmv_sort_tiles = 0;   // plan copies: entries of a tile sorted by column (measured 10 % SLOWER: profiles/r01v)
    int64_t spmv_relabel = 0;      // sliced plan: columns relabelled by count class, x permuted per SpMV: 0 auto (on), 1 on, 2 off
    int64_t spmv_tile = 0;         // nnz per workgroup tile: 0 auto, 2048 or 4096
    int64_t spgemm_task_order = 0; // large-row tasks: 0/1 costliest first (stable sort by cost class), 2 row order (A/B)
    int64_t spgemm_xcd_chunk = 0;  // large-row task list -> XCDs: 0 round-robin, -1 one contiguous run per XCD
    int64_t spgemm_bucket = 1;     // SpGEMM: column-bucket table of B instead of binary searches (A/B)
    int64_t spgemm_prof = 0;       // SpGEMM: print the time of the large-row numeric tasks by class and the longest ones (debug)
    int64_t spgemm_tokens = 1;     // SpGEMM: token chains of the workgroup kernel's ordered adds (accumulators split by rank mod n): 1, 2 or 4
    int64_t spgemm_overlap = 0;    // SpGEMM: wave kernels (hash rows, wave-per-row rows) on a second stream beside the large-row kernel
    int64_t spgemm_midwin_sym = 16; // SpGEMM: log2 of the window of the wave-per-row COUNTING kernel (14..16)
    int64_t spgemm_midwin = 14;    // SpGEMM: log2 of the column window of the wave-per-row kernel (13 or 14)
    int64_t spgemm_mid = 65536;    // SpGEMM: rows of <= 64 k's and at most this many products run one wave per row (0: none)
    int64_t spgemm_debug = 0;      // SpGEMM TIMING EXPERIMENTS ONLY (wrong results): 1 no ordering of the adds, 2 no index emission
    int64_t spgemm_occupancy = 3;  // SpGEMM: workgroups per CU the large-row numeric kernel is compiled for: 3 (80 VGPRs) or 2 (128 VGPRs) (A/B)
    int64_t spgemm_retain = 1;     // SpGEMM: windows of few entries keep them in registers from the bit pass to the adds (A/B)
    int64_t spgemm_lds_atomic = 1; // SpGEMM: value adds as ds_add_f64 (1) or read / add / write (0); same order either way (A/B)
    int64_t spgemm_winlog = 17;    // SpGEMM: log2 of the widest column window of a large-row task (16..19)
    int64_t spgemm_minwin = 13;    // SpGEMM: log2 of the narrowest column window of a heavy row (11..16)
    int64_t spgemm_heavy = 131072;    // SpGEMM: a row of more products is cut into one task per (narrower) column window, about this many products each
    int64_t pool = 1;              // keep released result blocks (>= 1 MiB) for the next result instead of hipFree
    int64_t pool_max_bytes = 128ll << 30;   // cap on the bytes the pool may hold
    int64_t spmv_band = 0;         // banded plan (hot columns from LDS, spmv_band.hip) instead of the XCD-sliced one: 0 auto (on), 1 on, 2 off
    int64_t spmv_band_hot = 0;     // hot slices of 8192 labels each (0 = default 128)
    int64_t spmv_band_phases = 0;  // label ranges of the cold rest (0 = default 1), 8 hash pieces each
    int64_t spmv_band_split_launch = 0;   // profiling: short rows and cold pieces in two launches instead of one
    int64_t spmv_band_hot_threads = 0;    // threads per workgroup of the hot kernel: 1024 (default) or 512
    int64_t spmv_band_gather = 0;         // how the cold kernel reads x: 0 plain, 1 non-temporal, 2 device scope (L1 bypass)
    int64_t spmv_band_overlap = 0;        // cold pieces + short rows on a second stream beside the hot kernel: 0/1 on, 2 off
    int64_t spmv_band_natural = 0;        // cold entries keep their original column and read x itself (no per-SpMV scatter of x): 1 on (measured slower: profiles/r03i), 0/2 off
    int64_t spmv_band_split_permute = 0;  // with the overlap: hot labels of x gathered first, the rest scattered on the second stream: 0/1 on, 2 off
    int64_t spmv_band_short = 0;          // short rows: 0/2 as one more gather piece, 1 tiled with the 8192 hottest x entries in LDS (measured slower)
    int64_t spmv_band_short_group = 0;    // blocks per workgroup of the tiled short-rows launch (0 = default 4)
    int64_t spmv_band_split = 0;          // rows with at least this many entries are cut into pieces (0 = default 24)
    int64_t spmv_band_group = 0;   // blocks of 8192 entries per workgroup of the hot kernel (0 = default 16)
    int64_t spmm_long_row = -1;    // SpMM: rows of more entries go to the chunk kernels (summation by chunks); -1 = default (0: all rows); L > 0: rows of <= L entries are summed in entry order by lane groups (the reference's bits, 3x slower)
    int64_t spmv_lds_pad = 0;      // extra dynamic LDS bytes per workgroup: caps workgroups per CU (tuning)
    int64_t spmv_xmask = -1;       // TIMING EXPERIMENTS ONLY: gather x[col & mask] (wrong results unless -1)
};
Options &options();

// ---- SpMV plan ---------------------------------------------------------------
constexpr int XCS_SLICES = 8;      // one slice of x lines per XCD (each XCD has a private 4 MiB L2)

// One CSR piece the tile kernel runs over.
struct CsrPiece {
    void *indptr = nullptr;        // device; PTR of the handle for `main`, uint64 for slices
    void *indices = nullptr;       // device
    double *data = nullptr;        // device
    uint16_t *pos = nullptr;       // device, plan copies only: tile-local origin of each entry (tiles sorted by column)
    uint64_t rows = 0, nnz = 0, ntiles = 0;
    uint64_t *tile_row = nullptr;  // device, ntiles + 1: first row starting at/after tile c
    bool owns = false;             // arrays allocated by the plan (copies), not borrowed from the handle
};

struct SpmvScratch {               // per stream: nothing in here is shared between in-flight SpMVs
    double *carry_main = nullptr;
    double *carry_slices = nullptr;
    double *partial = nullptr;     // XCS_SLICES x n_long partial row sums
    double *xp = nullptr;          // x in the plan's column labelling (relabelled plans)
};

struct BandPlan;                   // spmv_band.hip

struct SpmvPlan {
    bool built = false;
    bool xcs = false;
    BandPlan *band = nullptr;      // banded plan: when set, nothing else below is used
    int64_t opt_band = -1, opt_band_hot = -1, opt_band_phases = -1, opt_band_group = -1, opt_band_split = -1, opt_band_natural = -1, opt_band_short = -1;
    int64_t opt_xcs = -1, opt_split = -1, opt_idx32 = -1, opt_tile = -1, opt_sort = -1, opt_relabel = -1;   // option values the plan was built with
    uint32_t tile = 0;             // nnz per tile
    int idx_bytes = 8;             // width of the column ids the kernels read (handle's, or 4 for plan copies)
    CsrPiece main;                 // the whole matrix (plain plan) or its short rows (sliced plan)
    CsrPiece slice[XCS_SLICES];    // long rows, entries whose x line hashes to s, rows = n_long
    uint64_t slice_tile_off[XCS_SLICES + 1] = {0};
    uint64_t n_long = 0;
    uint64_t *long_rows = nullptr; // device: original row of long row j
    uint32_t *perm = nullptr;      // device, cols entries: plan label of column j (relabelled plans), else null
    uint64_t cols = 0;
    void *slab = nullptr;          // backing store of all plan-owned arrays
    std::unordered_map<void *, SpmvScratch> scratch;
    void release();
};

int32_t exclusive_scan_u64(const uint64_t *in, uint64_t *out, uint64_t n, hipStream_t stream);   // scan.hip

// ---- SpMM plan: rows cut into chunks of <= 512 entries (spmm.hip) -------------------------------
struct SpmmPlan {
    bool built = false;
    uint64_t nchunks = 0, n_multi = 0, long_row = 0;
    uint64_t *first_chunk = nullptr;   // device, rows + 1
    uint64_t *chunk_row = nullptr;     // device, nchunks
    uint64_t *multi_rows = nullptr;    // device, rows spanning several chunks
    std::unordered_map<void *, std::pair<double *, uint64_t>> partial;   // per stream: buffer, bytes
    void release();
};

}  // namespace sprs_hip

// Device twin of CsMatBase (sprs/src/sparse.rs:94-122).
struct sprs_hip_csmat {
    int32_t storage = SPRS_HIP_CSR;
    uint64_t rows = 0, cols = 0, nnz = 0;
    int32_t iptr_bytes = 8, idx_bytes = 8;       // widths of the device arrays (4 or 8)
    // widths the CALLER declared (2, 4 or 8): sprs' SpIndex covers u16 / i16 too (indexing.rs:124-130).  2-byte arrays are
    // widened to 4 bytes on upload and narrowed on download; every value a result could hold is checked against the
    // declared width where the reference's I::from_usize / Iptr::from_usize would panic.  0 = same as the device width.
    int32_t decl_iptr_bytes = 0, decl_idx_bytes = 0;
    int32_t user_iptr_bytes() const { return decl_iptr_bytes ? decl_iptr_bytes : iptr_bytes; }
    int32_t user_idx_bytes() const { return decl_idx_bytes ? decl_idx_bytes : idx_bytes; }
    void *indptr = nullptr;    // device, outer+1 entries, zero based
    void *indices = nullptr;   // device, nnz entries
    double *data = nullptr;    // device, nnz entries
    bool owns = false;
    uint64_t cap_indices = 0, cap_data = 0;   // bytes of the owned blocks (>= what nnz needs: blocks come from the pool)
    int device = 0;
    std::recursive_mutex mu;   // guards plan / mm: held from the look-up (or rebuild) of a plan until the kernels that read it are launched
    sprs_hip::SpmvPlan plan;
    sprs_hip::SpmmPlan mm;

    uint64_t outer() const { return storage == SPRS_HIP_CSR ? rows : cols; }
    uint64_t inner() const { return storage == SPRS_HIP_CSR ? cols : rows; }
};

namespace sprs_hip {

// abi.hip: result-block pool
hipError_t pool_alloc(void **p, uint64_t bytes, uint64_t *cap, int device);
void pool_free(void *p, uint64_t cap, int device);
uint64_t pool_trim();
uint64_t pool_cached_bytes();

// spmv_band.hip
int32_t band_build(sprs_hip_csmat *a, hipStream_t stream, BandPlan **out);   // *out stays null when the plan does not apply
int32_t band_spmv(sprs_hip_csmat *a, BandPlan *bp, const double *x, double *y, bool accumulate, hipStream_t stream);
void band_free(BandPlan *bp);
uint64_t band_plan_bytes(const BandPlan *bp);
// spmv.hip
int32_t spmv_f64(sprs_hip_csmat *a, const double *x, double *y, bool accumulate, hipStream_t stream);
// spgemm.hip
int32_t spgemm_f64(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c);
int32_t spgemm_symbolic(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c);
int32_t spgemm_numeric(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c);
int32_t spgemm_plan_create(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_spgemm_plan **out);
int32_t spgemm_plan_structure(sprs_hip_spgemm_plan *pl, const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **out, bool with_values);
int32_t spgemm_plan_numeric(sprs_hip_spgemm_plan *pl, const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c);
uint64_t spgemm_plan_nnz(const sprs_hip_spgemm_plan *pl);
void spgemm_plan_free(sprs_hip_spgemm_plan *pl);
// spmm.hip
int32_t spmm_rowmaj_f64(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t ld_rhs, double *out,
                        uint64_t ld_out, bool accumulate, hipStream_t stream);
// dist.hip
int32_t dist_unique_id(void *id128);
int32_t dist_create(sprs_hip_dist **out, const void *unique_id128, int32_t world, int32_t rank, uint64_t rows, uint64_t cols,
                    const uint64_t *row_starts, const sprs_hip_csmat *local_block, int32_t nsub);
int32_t dist_spmv(sprs_hip_dist *d, const double *x, double *y, hipStream_t stream);
void dist_free(sprs_hip_dist *d);
uint64_t dist_rows(const sprs_hip_dist *d);
uint64_t dist_cols(const sprs_hip_dist *d);
// triplet.hip
int32_t triplets_to_cs(uint64_t rows, uint64_t cols, uint64_t n, const void *row_inds, const void *col_inds, int32_t in_idx_bytes,
                       const double *data, int32_t storage, int32_t out_idx_bytes, int32_t out_iptr_bytes, sprs_hip_csmat **out);
// convert.hip
int32_t to_other_storage(const sprs_hip_csmat *m, sprs_hip_csmat **out);
// bicgstab.hip
int32_t bicgstab_f64(sprs_hip_csmat *a, const double *x0, const double *b, uint64_t n, double tol, uint64_t max_iter,
                     double soft_restart_threshold, double *x, sprs_hip_bicgstab_info *info, hipStream_t stream);
int32_t slice_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end, sprs_hip_csmat **out);
// abi.hip
int32_t alloc_csmat(sprs_hip_csmat **out, int32_t storage, uint64_t rows, uint64_t cols, uint64_t nnz,
                    int32_t iptr_bytes, int32_t idx_bytes);
// a result inherits the declared index widths of its operand; INDEX_OVERFLOW where a value would not fit them
int32_t inherit_declared_widths(sprs_hip_csmat *result, const sprs_hip_csmat *from);

}  // namespace sprs_hip
