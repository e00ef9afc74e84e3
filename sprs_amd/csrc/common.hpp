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
// One table for the struct, sprs_hip_set_option and sprs_hip_get_option: X(name, default, min, max, devtools).
// devtools = 1: the option only exists in builds with -DSPRS_HIP_DEVTOOLS (timing experiments that give WRONG results,
// profiling printouts); the release library rejects it with INVALID_ARG and compiles the code behind it out.
#ifdef SPRS_HIP_DEVTOOLS
constexpr bool DEVTOOLS = true;
#else
constexpr bool DEVTOOLS = false;
#endif
#define SPRS_HIP_OPTIONS(X)                                                                                                     \
    X(spmv_kernel, 0, 0, 2, 0)          /* 0 auto, 1 nnz-tiled, 2 wave-per-row (A/B only) */                                     \
    X(spmv_xcs, 0, 0, 2, 0)             /* XCD-sliced plan for long rows: 0 auto, 1 force on, 2 off */                            \
    X(spmv_xcs_split, 32, 2, INT64_MAX, 0) /* rows with >= this many entries go to the sliced part */                             \
    X(spmv_xcs_idx32, 1, 0, 1, 0)       /* plan-owned copies store 32-bit column ids when cols < 2^32 */                          \
    X(spmv_sort_tiles, 0, 0, 1, 0)      /* plan copies: entries of a tile sorted by column (10 % SLOWER: profiles/r01v) */         \
    X(spmv_relabel, 0, 0, 2, 0)         /* sliced plan: columns relabelled by count class: 0 auto (on), 1 on, 2 off */             \
    X(spmv_tile, 0, 0, 4096, 0)         /* nnz per workgroup tile: 0 auto, 2048 or 4096 */                                        \
    X(spmv_lds_pad, 0, 0, 100000, 0)    /* extra dynamic LDS bytes per workgroup: caps workgroups per CU (tuning) */              \
    X(spmv_xmask, -1, INT64_MIN, INT64_MAX, 1) /* TIMING EXPERIMENTS ONLY: gather x[col & mask] (wrong results unless -1) */      \
    X(spmv_plan_defer, 1, 0, 1, 0)      /* 1: a handle's FIRST multiply runs on the plain tile index; the re-laid-out copy plans (banded, XCD-sliced: ~0.1 s to build on a 3e8-entry matrix = 80 SpMVs) are built at the second one, or by sprs_hip_csmat_prepare; 0: at the first (rounds 1-4).  Forced plans (spmv_band = 1, spmv_xcs = 1) are never deferred */ \
    X(spmv_band, 0, 0, 2, 0)            /* banded plan (hot columns from LDS, spmv_band.hip): 0 auto (on), 1 on, 2 off */          \
    X(spmv_band_hot, 0, 0, 384, 0)      /* hot slices (0 = default 128) */                                                        \
    X(spmv_band_tile, 0, 0, 16384, 0)   /* labels per hot slice = doubles of the x tile in LDS: 8192 or 16384 (0 = default 16384) */ \
    X(spmv_band_phases, 0, 0, 8, 0)     /* label ranges of the cold rest (0 = default 1), 8 hash pieces each */                   \
    X(spmv_band_split, 0, 0, INT64_MAX, 0) /* rows with at least this many entries are cut into pieces (0 = default 24) */        \
    X(spmv_band_rounds, 0, 0, 64, 0)    /* workgroups of the hot kernel per CU, one after the other (0 = default 2) */ \
    X(spmv_band_debug, 0, 0, 255, 1)    /* TIMING EXPERIMENTS ONLY (wrong results): hot kernel 1 no stores of the row sums, 2 one row start per lane, 4 none */ \
    X(spmv_band_tail, 0, 0, 2, 0)       /* with the overlap: 0/1 the reduction of the long rows starts when the hot slices and the cold pieces are done, beside what is left of the short rows; 2 it waits for the whole second stream (rounds 2-4, A/B); on ONE stream (small plans): 0/1 the short rows share the reduction's launch (band_tail_kernel), 2 they run as their own launch in front of the hot slices (round 5) */ \
    X(spmv_band_balance, 0, 0, 2, 0)    /* shares of the hot workgroups: by modelled COST of their tiles (1 + row ends / 180) — 0 auto (small plans, whose hot kernel runs alone on one stream), 1 on, 2 off (equal tile counts) */ \
    X(spmv_band_xcd, 0, 0, 2, 0)        /* hot workgroups: every XCD takes a contiguous run of shares (its L2 serves a slice's x tile to its neighbours) — 0 auto (small plans), 1 on, 2 off (shares dealt round-robin over the XCDs) */ \
    X(spmv_band_share, 0, 0, 1ll << 30, 0) /* wave tiles per workgroup of the hot kernel (0 = default: an equal share for rounds x CUs workgroups) */ \
    X(spmv_band_hot_run, 0, 0, 4096, 0) /* consecutive wave tiles per range of the hot kernel (0 = default 4); a workgroup streams 16 neighbouring ranges */ \
    X(spmv_band_cold_tiles, 0, 0, 64, 0) /* consecutive wave tiles per wave of the cold kernel (0 = default 4) */                 \
    X(spmv_band_overlap, 0, 0, 2, 0)    /* cold pieces + short rows on a second stream beside the hot kernel: 0/1 on, 2 off */     \
    X(spmv_band_split_permute, 0, 0, 2, 0) /* with the overlap: hot labels of x gathered first, the rest scattered on the second stream: 0/1 on, 2 off */ \
    X(spmm_long_row, -1, -1, INT64_MAX, 0) /* SpMM: -1 / 0 default (the entry-stream kernel); L > 0: rows of <= L entries summed in entry order by lane groups, longer ones by 512-entry chunks (reference bits for the short rows, several times slower) */ \
    X(spmm_stream, 1, 0, 1, 0)          /* SpMM: 1 tiles of 256 consecutive entries per wave (default); 0 one wave per row chunk (the kernel of rounds 1-3; also what a matrix with 2^32 or more columns runs) */ \
    X(spgemm_micro, 0, 0, 2, 0)         /* rows of at most 64 products and k's by lane groups, 4 / 2 / 1 rows per wave, no LDS (micro_rows_kernel): 0/1 on, 2 off (the hash kernel, A/B) */ \
    X(spgemm_task_order, 0, 0, 2, 0)    /* large-row tasks: 0/1 costliest first (stable sort by cost class), 2 row order (A/B) */  \
    X(spgemm_xcd_chunk, 0, -1, 0, 0)    /* large-row task list -> XCDs: 0 round-robin, -1 one contiguous run per XCD */            \
    X(spgemm_bucket, 1, 0, 1, 0)        /* column-bucket table of B instead of binary searches (A/B) */                           \
    X(spgemm_prof, 0, 0, 1, 1)          /* print the time of the numeric tasks by class and the longest ones */                   \
    X(spgemm_tokens, 1, 1, 4, 0)        /* token chains of the workgroup kernel's ordered adds: 1, 2 or 4 */                      \
    X(spgemm_keep_bits, 1, 0, 1, 0)     /* the counting pass keeps the bitmaps of the large rows for the numeric pass (one walk of their entries instead of two); 0: the numeric kernel walks for its bits (A/B) */ \
    X(spgemm_lane_order, 0, 0, 2, 0)    /* products of one wave instruction into the LDS accumulators: 0 auto = ONE ds_add_f64 when the device passes the lane-order probe (same-address lanes applied in ascending lane order), else one instruction per k-run; 2 always per k-run (A/B); same bits either way */ \
    X(spgemm_overlap, 0, 0, 1, 0)       /* wave kernels on a second stream beside the large-row kernel */                         \
    X(spgemm_midwin_sym, 14, 14, 16, 0) /* log2 of the window of the wave-per-row COUNTING kernel: 14 or 16 */                        \
    X(spgemm_mid_keep, 8, 4, 8, 0)      /* wave instructions per chunk of the wave-per-row kernel (loads in flight together; kept in registers through a segment): 4 or 8 */ \
    X(spgemm_mid_keep_sym, 8, 8, 16, 0) /* the same for its counting twin: 8 or 16 */ \
    X(spgemm_midwin, 15, 14, 15, 0)     /* log2 of the column window of the wave-per-row kernel: 14 or 15 (measured equal on config 5; 2^16 — 10 waves per CU — slower: profiles/r10d) */ \
    X(spgemm_mid, 65536, 0, 1ll << 31, 0) /* rows of <= 64 k's and at most this many products run one wave per row (0: none) */   \
    X(spgemm_ordered, 1, 0, 1, 0)       /* 1: products are added in the reference's order (values bit-identical to sprs'); 0: the waves of a large-row workgroup add as they arrive (LDS atomics: same products, rounding-level differences, not reproducible run to run; ~20 % faster kernel) */ \
    X(spmm_relayout, 0, 0, 2, 0)        /* SpMM stream kernel: gather from a copy of the rhs whose rows are scattered by a multiplicative hash inside blocks of 4096 rows (hub columns sit at 0, 2^k, 2^j + 2^k: their rhs rows would share a few L2 / fabric channels): 0 auto (a rhs of >= 256 MiB per column block under >= 24 entries per column whose row pitch is a power of two or leaves rows of < 16 columns across lines, or one that is not row-major), 1 on, 2 off */ \
    X(spmm_debug, 0, 0, 7, 1)           /* developer A/B of the stream kernel (right results): 1 plain instead of non-temporal entry loads, 2 plain result stores, 4 non-temporal rhs gathers */ \
    X(spgemm_debug, 0, 0, 15, 1)        /* TIMING EXPERIMENTS ONLY (wrong results): 1 no ordering of the adds, 2 no index emission, 4 no value stores and 8 no adds (wave-per-row kernel) */ \
    X(spgemm_occupancy, 3, 2, 3, 0)     /* workgroups per CU the large-row numeric kernel is compiled for: 3 (80 VGPRs) or 2 (128) */ \
    X(spgemm_lds_atomic, 1, 0, 1, 0)    /* value adds as ds_add_f64 (1) or read / add / write (0); same order either way (A/B) */  \
    X(spgemm_winlog, 17, 16, 19, 0)     /* log2 of the widest column window of a large-row task */                                \
    X(spgemm_minwin, 13, 11, 16, 0)     /* log2 of the narrowest column window of a heavy row */                                  \
    X(spgemm_heavy, 524288, 1024, INT64_MAX, 0) /* a row of more products is cut into one task per (narrower) column window */    \
    X(gauss_seidel_blocks, 0, 0, 65536, 0) /* workgroups (4 waves) of the Gauss-Seidel sweep kernel (0 = default: one per CU) */          \
    X(gauss_seidel_chain, 0, 0, 1ll << 30, 0) /* 0 / 1: rows in dependency-level order, one row per lane; S > 1: the band schedule — chains of S consecutive rows per lane, skewed, hand-offs through LDS inside a workgroup (bit-identical; measured not faster as built: DESIGN 4.5) — with this stride, INVALID_ARG when the matrix does not fit it */ \
    X(gauss_seidel_debug, 0, 0, 255, 1) /* TIMING EXPERIMENTS ONLY (wrong results): band kernel without 1 any pipeline work, 2 operand loads, 4 entry loads, 8 index loads */ \
    X(gauss_seidel_xcd, 0, 0, 2, 0)     /* sweep kernel: 1 only the workgroups that find themselves on XCD 0 take part (hand-offs through ONE L2), 0 / 2 every XCD (measured: one XCD is not faster) */ \
    X(gauss_seidel_naps, 0, 0, 64, 0)   /* longest pause of a wave whose rows all wait, in s_sleep(1) units, growing with the wait (0 = default 1) */ \
    X(pool, 1, 0, 1, 0)                 /* keep released result blocks (>= 1 MiB) for the next result instead of hipFree */        \
    X(pool_max_bytes, 128ll << 30, 0, INT64_MAX, 0) /* cap on the bytes the pool may hold */

struct Options {
#define SPRS_X(name, def, lo, hi, dev) int64_t name = (def);
    SPRS_HIP_OPTIONS(SPRS_X)
#undef SPRS_X
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
    bool light = false;            // built as the plain tile index although a copy plan would apply (first multiply of the handle): rebuilt at the next one
    bool xcs = false;
    BandPlan *band = nullptr;      // banded plan: when set, nothing else below is used
    uint64_t opt_sig = 0;          // hash of the option values the plan was built with (plan_signature, spmv.hip)
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
    uint64_t *tile_row = nullptr;      // device, ntiles + 1: the row that holds entry t * 256 (entry-stream kernel)
    uint64_t ntiles = 0;
    bool stream = false;               // built for the entry-stream kernel (no chunk lists then)
    std::unordered_map<void *, std::pair<double *, uint64_t>> partial;   // per stream: buffer, bytes
    std::unordered_map<void *, std::pair<double *, uint64_t>> relaid;    // per stream: the re-laid-out copy of the rhs (option spmm_relayout)
    void release();
};

// ---- Gauss-Seidel plan: the rows in dependency-level order (gauss_seidel.hip) ---------------------
struct GsPlan {
    bool built = false;
    uint32_t *order = nullptr;         // device, rows entries: row swept at position q (levels ascending, rows ascending inside a level)
    uint64_t nlevels = 0;
    uint64_t no_diag_row = UINT64_MAX; // first row without a stored diagonal entry (the reference's diag.unwrap() panics there)
    uint64_t chain_tried = 0;          // stride the band schedule was last checked for (0: never) ...
    bool chain_ok = false;             // ... and whether every dependency fits it (gs_band_check_kernel)
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
    uint64_t spmv_calls = 0;   // multiplies so far: the re-laid-out plan copies (banded / XCD-sliced) are built at the SECOND one (or by sprs_hip_csmat_prepare)
    bool prepared = false;     // sprs_hip_csmat_prepare was called: the copy plans may be built at once
    bool one_shot = false;     // the handle multiplies once (sprs_hip_spmv_f64_host): plain plan, no copies of the matrix
    uint64_t cap_indptr = 0, cap_indices = 0, cap_data = 0;   // bytes of the owned blocks (>= what nnz needs: blocks come from the pool)
    int device = 0;
    std::recursive_mutex mu;   // guards plan / mm: held from the look-up (or rebuild) of a plan until the kernels that read it are launched
    sprs_hip::SpmvPlan plan;
    sprs_hip::SpmmPlan mm;
    sprs_hip::GsPlan gs;
    sprs_hip_csmat *t_view = nullptr;    // transpose view (of the CSC form) kept for dense . sparse products (sprs_hip_dense_dot_csmat_f64): its SpMM / SpMV plans live as long as the values do; dropped with as_other
    sprs_hip_csmat *as_other = nullptr;  // the handle in the OTHER storage order (to_other_storage, csmat.rs:1405-1426): made by the first product that needs it (a CSC operand of a dense product / SpMV runs on its CSR form), dropped by refresh / free

    uint64_t outer() const { return storage == SPRS_HIP_CSR ? rows : cols; }
    uint64_t inner() const { return storage == SPRS_HIP_CSR ? cols : rows; }
};

namespace sprs_hip {

// abi.hip: result-block pool
hipError_t pool_alloc(void **p, uint64_t bytes, uint64_t *cap, int device);
// stream_ordered = true: the block was only ever touched by work enqueued on the NULL stream (or on streams joined back to it)
// and every later user of the pool enqueues there too, so it goes back without waiting for the device
void pool_free(void *p, uint64_t cap, int device, bool stream_ordered = false);
uint64_t pool_trim();
uint64_t pool_cached_bytes();

// spmv_band.hip
int32_t band_build(sprs_hip_csmat *a, hipStream_t stream, BandPlan **out);   // *out stays null when the plan does not apply
int32_t band_spmv(sprs_hip_csmat *a, BandPlan *bp, const double *x, double *y, bool accumulate, hipStream_t stream);
void band_free(BandPlan *bp);
uint64_t band_plan_bytes(const BandPlan *bp);
// spmv.hip
int32_t spmv_prepare(sprs_hip_csmat *a, hipStream_t stream);   // builds the full SpMV plan now (sprs_hip_csmat_prepare)
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
int32_t spmm_strided_f64(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t rs_rhs, uint64_t cs_rhs, double *out,
                         uint64_t rs_out, uint64_t cs_out, bool accumulate, hipStream_t stream);
// dist.hip
int32_t dist_unique_id(void *id128);
int32_t dist_create(sprs_hip_dist **out, const void *unique_id128, int32_t world, int32_t rank, uint64_t rows, uint64_t cols,
                    const uint64_t *row_starts, const sprs_hip_csmat *local_block, int32_t nsub);
int32_t dist_spmv(sprs_hip_dist *d, const double *x, double *y, hipStream_t stream);
void dist_free(sprs_hip_dist *d);
uint64_t dist_rows(const sprs_hip_dist *d);
uint64_t dist_cols(const sprs_hip_dist *d);
int32_t dist_comm_count(const sprs_hip_dist *d, int32_t *ranks);
int32_t dist_peer_handle(sprs_hip_dist *d, void *handle64);                       // the peer-store route (dist.hip)
int32_t dist_peer_connect(sprs_hip_dist *d, const void *handles, int32_t world);
int32_t dist_set_route(sprs_hip_dist *d, int32_t route);
int32_t dist_route(const sprs_hip_dist *d, int32_t *route);
// triplet.hip
int32_t triplets_to_cs(uint64_t rows, uint64_t cols, uint64_t n, const void *row_inds, const void *col_inds, int32_t in_idx_bytes,
                       const double *data, int32_t storage, int32_t out_idx_bytes, int32_t out_iptr_bytes, sprs_hip_csmat **out);
// convert.hip
int32_t to_other_storage(const sprs_hip_csmat *m, sprs_hip_csmat **out);
// bicgstab.hip
int32_t bicgstab_f64(sprs_hip_csmat *a, const double *x0, const double *b, uint64_t n, double tol, uint64_t max_iter,
                     double soft_restart_threshold, double *x, sprs_hip_bicgstab_info *info, hipStream_t stream);
// gauss_seidel.hip
int32_t gauss_seidel_f64(sprs_hip_csmat *a, double *x, const double *rhs, uint64_t n, uint64_t max_iter, double eps,
                         sprs_hip_gauss_seidel_info *info, hipStream_t stream);
int32_t slice_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end, sprs_hip_csmat **out);
// abi.hip
int32_t alloc_csmat(sprs_hip_csmat **out, int32_t storage, uint64_t rows, uint64_t cols, uint64_t nnz,
                    int32_t iptr_bytes, int32_t idx_bytes);
// a result inherits the declared index widths of its operand; INDEX_OVERFLOW where a value would not fit them
int32_t inherit_declared_widths(sprs_hip_csmat *result, const sprs_hip_csmat *from);

}  // namespace sprs_hip
