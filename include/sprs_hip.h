/*
 * sprs_hip.h — C ABI of libsprs_hip.so, the MI355X (gfx950) backend for the
 * sprs CSR SpMV / SpGEMM hot path.
 *
 * This is the boundary a `sprs-hip-sys` Rust crate binds (rust/sprs-hip-sys,
 * following the reference's own *_sys + safe-wrapper split,
 * suitesparse_bindings/suitesparse_ldl_sys/src/lib.rs:10-80 and
 * sprs_suitesparse_ldl/src/lib.rs:53-129).  Conventions, all taken from how
 * the reference already crosses a C ABI:
 *   - CSR is passed as (rows, cols, indptr*, indices*, data*) raw pointers
 *     (sprs-benches/src/main.rs:28-41  <->  sprs-benches/src/eigen.cpp:6-29);
 *   - index widths are given in bytes: 8 = usize/u64/i64/isize (sprs default,
 *     sprs/src/sparse.rs:111-122), 4 = u32/i32 (sprs/src/indexing.rs:124-130);
 *   - matrices are opaque handles freed by an explicit call, owned on the
 *     Rust side by a struct with Drop (sprs_suitesparse_umfpack/src/lib.rs:30-46);
 *   - every entry point returns a status; nothing unwinds across the boundary.
 *     Status codes map 1:1 onto the reference's panics / errors so the wrapper
 *     can re-raise them with the same text (see each code).
 *
 * All citations are relative to the sprs repository root.
 * No torch / C++ types appear here: plain pointers and sizes only.
 */
#ifndef SPRS_HIP_H
#define SPRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status ----------------------------------------------------------- */

#define SPRS_HIP_OK 0
/* assert "Dimension mismatch": prod.rs:114-117, prod.rs:283-285; assert_eq!(lhs.cols(), rhs.rows()) smmp.rs:207 */
#define SPRS_HIP_DIM_MISMATCH 1
/* assert "Storage mismatch": prod.rs:118, prod.rs:286 */
#define SPRS_HIP_STORAGE_MISMATCH 2
/* panic "Index type is not large enough to hold ...": csmat.rs:1794-1797; I::from_usize / Iptr::from_usize smmp.rs:116,121 */
#define SPRS_HIP_INDEX_OVERFLOW 3
/* StructureError::{Unsorted,SizeMismatch,OutOfRange}: errors.rs:4-8, sparse.rs:300-358 */
#define SPRS_HIP_BAD_STRUCTURE 4
/* null pointer / unsupported width / misaligned device buffer */
#define SPRS_HIP_INVALID_ARG 5
/* hipMalloc failed */
#define SPRS_HIP_OUT_OF_MEMORY 6
/* any other HIP runtime error -> LinalgError::ThirdPartyError(code, msg), errors.rs:70,94-96 */
#define SPRS_HIP_HIP_ERROR 7
/* no usable gfx950 device: the product path never falls back to the CPU */
#define SPRS_HIP_NO_DEVICE 8

/* CompressedStorage, sprs/src/sparse.rs:30-40 */
#define SPRS_HIP_CSR 0
#define SPRS_HIP_CSC 1
/* dense operands: C order (ndarray's standard layout) or Fortran order (`.f()`), with a leading dimension */
#define SPRS_HIP_ROW_MAJOR 0
#define SPRS_HIP_COL_MAJOR 1

/* Thread-local text of the last failure on the calling thread ("" if none),
 * and the raw hipError_t behind a SPRS_HIP_HIP_ERROR (0 otherwise). */
const char *sprs_hip_last_error(void);
int32_t sprs_hip_last_hip_code(void);
/* "sprs_hip <version> gfx950 ..." */
const char *sprs_hip_version(void);

/* ---- device / raw buffers (DeviceVec plumbing) ------------------------- */

int32_t sprs_hip_device_count(int32_t *count);
int32_t sprs_hip_set_device(int32_t device);
int32_t sprs_hip_malloc(void **dev_ptr, uint64_t bytes);
int32_t sprs_hip_free(void *dev_ptr);
int32_t sprs_hip_memcpy_h2d(void *dev_dst, const void *host_src, uint64_t bytes);
int32_t sprs_hip_memcpy_d2h(void *host_dst, const void *dev_src, uint64_t bytes);
int32_t sprs_hip_memcpy_d2d(void *dev_dst, const void *dev_src, uint64_t bytes, void *stream);
int32_t sprs_hip_memset(void *dev_dst, int32_t byte_value, uint64_t bytes, void *stream);
int32_t sprs_hip_synchronize(void *stream); /* NULL = whole device */
/* Result matrices (sprs_hip_spgemm_f64, sprs_hip_csmat_to_other_storage, uploads) take their index/value
 * blocks from a pool of previously released ones and return them there on sprs_hip_csmat_free (options
 * "pool" = 1, "pool_max_bytes"): the driver needs seconds to re-issue tens of GB that were just freed.
 * sprs_hip_pool_trim hands everything cached back to the driver (Rust analogue: dropping the CsMat really
 * frees — call this where that matters, e.g. before another library needs the memory). */
int32_t sprs_hip_pool_trim(uint64_t *freed_bytes /* may be NULL */);

/* ---- device CSR/CSC container: twin of CsMatBase (sparse.rs:94-122) ---- */

typedef struct sprs_hip_csmat sprs_hip_csmat;

/* Copies a host matrix into device buffers OWNED by the handle (the host
 * arrays stay the caller's).  `indptr` has outer+1 entries and may be
 * non-zero-based, as slice_outer views are (indptr.rs:118-124, 216-219):
 * `indices`/`data` then point at the element addressed by indptr[0], and the
 * device copy is rebased (to_proper, indptr.rs:206-214).
 * validate != 0 runs check_compressed_structure (sparse.rs:300-358) first and
 * returns SPRS_HIP_BAD_STRUCTURE where CsMat::new would panic; validate == 0
 * is new_trusted / new_unchecked (csmat.rs:265-301). */
/* Index widths: 4 or 8 bytes on the device.  Host arrays may also be 2 bytes wide (sprs' u16 / i16, indexing.rs:124-130):
 * they are widened to 4 bytes on upload, sprs_hip_csmat_info keeps reporting 2, downloads narrow again, and every
 * result derived from such a matrix is checked against the 2-byte range where the reference's I::from_usize /
 * Iptr::from_usize would panic (SPRS_HIP_INDEX_OVERFLOW; sprs/tests/gh374.rs is reproduced literally). */
int32_t sprs_hip_csmat_upload(sprs_hip_csmat **out, int32_t storage, uint64_t rows, uint64_t cols,
                              const void *indptr, int32_t iptr_bytes, const void *indices,
                              int32_t idx_bytes, const double *data, int32_t validate);

/* Wraps buffers that already live on the current device (NOT owned, must
 * outlive the handle; zero-based indptr; 16-byte aligned). */
int32_t sprs_hip_csmat_wrap_device(sprs_hip_csmat **out, int32_t storage, uint64_t rows,
                                   uint64_t cols, uint64_t nnz, const void *dev_indptr,
                                   int32_t iptr_bytes, const void *dev_indices, int32_t idx_bytes,
                                   const double *dev_data);

/* shape / nnz / widths / storage; any out pointer may be NULL */
int32_t sprs_hip_csmat_info(const sprs_hip_csmat *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz,
                            int32_t *iptr_bytes, int32_t *idx_bytes, int32_t *storage);
/* raw device pointers (into_raw_storage, csmat.rs:946-954), still owned by the handle */
int32_t sprs_hip_csmat_device_ptrs(const sprs_hip_csmat *m, const void **indptr,
                                   const void **indices, const double **data);
/* whole matrix to host buffers of outer+1 / nnz / nnz entries */
int32_t sprs_hip_csmat_download(const sprs_hip_csmat *m, void *indptr, void *indices, double *data);
/* outer slices [start, end) only — slice_outer (slicing.rs:65-89).  indptr_out
 * gets end-start+1 NON-rebased entries (as the reference's view does);
 * indices_out/data_out get indptr[end]-indptr[start] entries.  Either may be
 * NULL to query sizes through *nnz_out. */
int32_t sprs_hip_csmat_download_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end,
                                      void *indptr_out, void *indices_out, double *data_out,
                                      uint64_t *nnz_out);
/* slice_outer (slicing.rs:65-89) as a NEW owning handle: outer slices [start, end), indptr
 * rebased (to_proper, indptr.rs:206-214).  The reference's view is zero-copy; the device twin
 * materialises the slice (one D2D copy) so that it is a self-contained, 16-byte aligned matrix —
 * this is how the multi-GPU path cuts row blocks. */
int32_t sprs_hip_csmat_slice_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end,
                                   sprs_hip_csmat **out);
/* Handles are immutable snapshots, like a `&CsMat`: the multiply plans cached inside a handle
 * (tile index, XCD-sliced copy, SpMM chunks) are derived from its arrays on first use.  After
 * modifying the arrays of a WRAPPED handle in place (sprs_hip_csmat_wrap_device), call this to drop
 * the cached plans; they are rebuilt by the next multiply. */
int32_t sprs_hip_csmat_refresh(sprs_hip_csmat *m);
/* PLAN POLICY.  A handle's FIRST SpMV runs on the plain tile index over its own arrays (built in microseconds); the plans
 * that re-lay the matrix out — the banded copy with the hot columns served from LDS, the XCD-sliced copy: ~0.1 s to build
 * for a 3e8-entry matrix, i.e. about 80 SpMVs, and 0.7 x the matrix in extra HBM — are built by the SECOND SpMV of the
 * handle, so a handle that multiplies once (`&a * &x` on a temporary) never pays for them.  A caller that knows it will
 * iterate (a solver) calls this once after the upload: the full plan is built now, on `stream`, and the first SpMV already
 * runs at the steady-state rate.  Idempotent; option spmv_plan_defer = 0 brings back "build at the first multiply".
 * What a caller that does NOT prepare must know: (i) multiply #1 and multiply #2 of one handle run on different plans and so
 * add the products of a row in different (each deterministic) orders — both within 1e-10 of the reference, but not
 * bit-identical to each other; from multiply #2 on every result has the same bits.  A solver whose first iteration must
 * reproduce later ones prepares the handle.  (ii) Multiply #2 allocates, builds and synchronises (the plan build): it must
 * not sit inside a timed or a CAPTURED region — an SpMV on a stream that is being captured into a hipGraph fails with
 * SPRS_HIP_INVALID_ARG unless the handle's final plan already exists.  The flag and the multiply count belong to the handle the
 * caller holds: a CSC handle hands them to its cached CSR copy, also to the one rebuilt after sprs_hip_csmat_refresh.
 * No counterpart in the reference (its CsMat has no derived state). */
int32_t sprs_hip_csmat_prepare(sprs_hip_csmat *m, void *stream);
/* What the SpMV plan cached in the handle looks like (after the first multiply; kind 0 before):
 * kind 1 = nnz tiles over the handle's own arrays, 2 = XCD-sliced copy (spmv.hip), 3 = banded copy
 * with the hot columns served from LDS (spmv_band.hip); plan_bytes = HBM the plan holds besides the
 * handle's arrays.  No counterpart in the reference (its CsMat has no derived state). */
int32_t sprs_hip_csmat_spmv_plan_info(const sprs_hip_csmat *m, int32_t *kind, uint64_t *plan_bytes);
/* transpose_view (csmat.rs:982-991): free, shares the buffers, flips storage + shape */
int32_t sprs_hip_csmat_transpose_view(const sprs_hip_csmat *m, sprs_hip_csmat **out);
int32_t sprs_hip_csmat_free(sprs_hip_csmat *m);

/* ---- SpMV --------------------------------------------------------------- */

/* Twin of prod::mul_acc_mat_vec_csr (prod.rs:103-127) when accumulate != 0:
 *     y[i] += sum_k A[i,k] * x[k]      (y keeps its previous content),
 * and of `&A * &x` (csmat.rs:2119-2160 -> prod::csr_mulacc_dense_colmaj,
 * prod.rs:274-298) when accumulate == 0:  y[i] = sum_k A[i,k] * x[k].
 * x_dev / y_dev are device pointers of x_len / y_len doubles.
 * SPRS_HIP_DIM_MISMATCH unless A.cols == x_len && A.rows == y_len;
 * SPRS_HIP_STORAGE_MISMATCH unless A is CSR.  Asynchronous on `stream`
 * (hipStream_t; NULL = default stream).  Products are formed with a separately
 * rounded multiply and add like MulAcc (mul_acc.rs:28-30); only the summation
 * ORDER inside a row differs from the reference (tree instead of left-to-right),
 * and it is run-to-run deterministic. */
int32_t sprs_hip_spmv_f64(const sprs_hip_csmat *a, const double *x_dev, uint64_t x_len,
                          double *y_dev, uint64_t y_len, int32_t accumulate, void *stream);

/* One-shot host form (upload, multiply, download, synchronous): what a plain
 * `prod::mul_acc_mat_vec_csr(mat.view(), &x[..], &mut y[..])` call maps to.  The handle lives for one multiply, so it
 * runs on the plain nnz-tiled plan: no re-laid-out copy of the matrix is built (the banded plan of a kept handle takes
 * ~0.1 s to build on a 3e8-entry matrix for a 1 ms SpMV). */
int32_t sprs_hip_spmv_f64_host(uint64_t rows, uint64_t cols, const void *indptr,
                               int32_t iptr_bytes, const void *indices, int32_t idx_bytes,
                               const double *data, const double *x, uint64_t x_len, double *y,
                               uint64_t y_len, int32_t accumulate);

/* ---- SpMM: CSR x dense row-major ------------------------------------------ */

/* Twin of prod::csr_mulacc_dense_rowmaj (prod.rs:189-214), the kernel `&CsMat * &Array2` uses for
 * a rhs of >= 8 columns (csmat.rs:2002-2016):  out[i, :] += A[i, c] * rhs[c, :]  over the stored
 * entries of row i (accumulate != 0), or the same on a zeroed `out` (accumulate == 0).
 * rhs_dev: rhs_rows x k doubles, row-major, leading dimension ld_rhs (>= k); out_dev: out_rows x k,
 * leading dimension ld_out.  SPRS_HIP_DIM_MISMATCH unless A.cols == rhs_rows && A.rows == out_rows;
 * SPRS_HIP_STORAGE_MISMATCH unless A is CSR.  Asynchronous on `stream`; deterministic. */
int32_t sprs_hip_spmm_rowmaj_f64(const sprs_hip_csmat *a, const double *rhs_dev, uint64_t rhs_rows,
                                 uint64_t k, uint64_t ld_rhs, double *out_dev, uint64_t out_rows,
                                 uint64_t ld_out, int32_t accumulate, void *stream);

/* ---- products with dense operands: the operator dispatch below the ABI --------------------------- */

/* Twin of prod::mul_acc_mat_vec_csc (prod.rs:74-99): y += A x for a CSC matrix.  SPRS_HIP_DIM_MISMATCH unless
 * A.cols == x_len && A.rows == y_len, then SPRS_HIP_STORAGE_MISMATCH unless A is CSC (prod.rs:88-92).  The reference scatters
 * column by column; here the matrix is converted ONCE to CSR on the device (to_other_storage, csmat.rs:1405-1426), the copy is
 * cached in the handle (dropped by sprs_hip_csmat_refresh / _free) and the CSR kernels run on it: every y[i] still receives
 * its products by ascending column. */
int32_t sprs_hip_mul_acc_mat_vec_csc_f64(const sprs_hip_csmat *a, const double *x_dev, uint64_t x_len, double *y_dev,
                                         uint64_t y_len, void *stream);

/* Twin of `&CsMat * &Array1` and its Dot (csmat.rs:2119-2178): y = A x on a zeroed result for EITHER storage —
 * CSR -> csr_mulacc_dense_colmaj, CSC -> csc_mulacc_dense_colmaj with one column (csmat.rs:2140-2156), the latter on the cached
 * CSR form.  SPRS_HIP_DIM_MISMATCH as the kernels' asserts (prod.rs:257-259, 284-289). */
int32_t sprs_hip_csmat_mul_vec_f64(const sprs_hip_csmat *a, const double *x_dev, uint64_t x_len, double *y_dev,
                                   uint64_t y_len, void *stream);

/* The four dense kernels of prod.rs in one entry: csr_mulacc_dense_rowmaj / _colmaj (prod.rs:189-214, 274-298) and
 * csc_mulacc_dense_rowmaj / _colmaj (prod.rs:219-270):  out += lhs * rhs  (accumulate != 0) or  out = lhs * rhs  on a zeroed
 * out (accumulate == 0), lhs in either storage (CSC through the cached CSR form), rhs (rhs_rows x k) and out (out_rows x k)
 * each in the layout its tag names with leading dimension ld (row-major: >= k, column-major: >= rows).  The reference's four
 * functions differ in the loop order only (every out[i, j] receives its products by ascending column index in all of them);
 * the layouts tell this entry how to address the operands.  Column-major x column-major with k < 8 — what `&CsMat * &Array2`
 * makes for fewer than 8 columns — runs column by column through the SpMV.  SPRS_HIP_DIM_MISMATCH unless lhs.cols == rhs_rows
 * && lhs.rows == out_rows (prod.rs:199-201, 228-230, 257-259, 284-289). */
int32_t sprs_hip_csmat_mulacc_dense_f64(const sprs_hip_csmat *lhs, const double *rhs_dev, uint64_t rhs_rows, uint64_t k,
                                        int32_t rhs_layout, uint64_t ld_rhs, double *out_dev, uint64_t out_rows,
                                        int32_t out_layout, uint64_t ld_out, int32_t accumulate, void *stream);

/* Twin of `&CsMat * &Array2` / `CsMat::dot(&Array2)` (csmat.rs:1989-2048, 2119-2137): the four arms (CSR | CSC) x (>= 8 columns |
 * fewer).  out_dev: lhs.rows x k doubles, contiguous; written in the layout the reference allocates — row-major for k >= 8,
 * column-major (`.f()`) below — which *out_layout reports. */
int32_t sprs_hip_csmat_mul_dense_f64(const sprs_hip_csmat *lhs, const double *rhs_dev, uint64_t rhs_rows, uint64_t k,
                                     int32_t rhs_layout, uint64_t ld_rhs, double *out_dev, int32_t *out_layout, void *stream);

/* Twin of `Array2::dot(&CsMat)` (csmat.rs:2050-2117): lhs (lhs_rows x lhs_cols, dense) . rhs (sparse) as (rhs^T lhs^T)^T with
 * the reference's free transposes (transpose_view, .t(), reversed_axes).  out_dev: lhs_rows x rhs.cols doubles, contiguous, in
 * the layout *out_layout reports (the reversed axes of what the transposed product allocates).  Synchronous (the transposed
 * operand is a temporary view).  A CSR rhs is converted once to CSC (cached in the handle). */
int32_t sprs_hip_dense_dot_csmat_f64(const double *lhs_dev, uint64_t lhs_rows, uint64_t lhs_cols, int32_t lhs_layout,
                                     uint64_t ld_lhs, const sprs_hip_csmat *rhs, double *out_dev, int32_t *out_layout,
                                     void *stream);

/* ---- BiCGSTAB: a caller that loops on the SpMV -------------- */

/* Counters and scalars of sprs::linalg::bicgstab::BiCGSTAB after solve()
 * (iteration_count / soft_restart_count / hard_restart_count / err / rho,
 * bicgstab.rs:236-262); converged = solve() returned Ok rather than Err. */
typedef struct sprs_hip_bicgstab_info {
    uint64_t iteration_count;
    uint64_t soft_restart_count;
    uint64_t hard_restart_count;
    double err;
    double rho;
    int32_t converged;
} sprs_hip_bicgstab_info;

/* Twin of BiCGSTAB::solve(a, x0, b, tol, max_iter) (bicgstab.rs:148-171) with device-resident dense
 * vectors: solves A x = b from the start vector x0, unpreconditioned, with the reference's soft restart
 * (|rho| / err^2 < soft_restart_threshold; the reference's default is 0.1, with_restart_threshold) and
 * hard restart (true residual recomputed before convergence is claimed).  A: square CSR or CSC handle
 * (a CSC operand is converted once on the device).  x0_dev, b_dev, x_dev: n doubles each; x_dev may not
 * alias x0_dev / b_dev.  Like the reference, running out of iterations is not an error: the status is
 * SPRS_HIP_OK, info->converged is 0 and x_dev holds the last iterate.  SPRS_HIP_DIM_MISMATCH unless
 * A.rows == A.cols == n.  Blocks until done (the restart decisions need the scalars on the host).
 * Element-wise arithmetic is bit-identical to the reference's; dot products are serial (== reference)
 * for n <= 2048 — the reference's own 4 x 4 test system reproduces bit for bit — and a fixed two-level
 * tree above; the SpMVs are sprs_hip_spmv_f64.  Deterministic run to run. */
int32_t sprs_hip_bicgstab_f64(sprs_hip_csmat *a, const double *x0_dev, const double *b_dev, uint64_t n,
                              double tol, uint64_t max_iter, double soft_restart_threshold, double *x_dev,
                              sprs_hip_bicgstab_info *info, void *stream);

/* ---- Gauss-Seidel: the other caller that loops on the SpMV -- */

/* What gauss_seidel() of the reference's heat example returns (sprs/examples/heat.rs:103-139):
 * Ok((it, error)) -> converged = 1, iterations = it (index of the sweep after which error < eps);
 * Err(error)      -> converged = 0, iterations = max_iter.  levels = length of the longest chain of
 * rows that must be swept one after the other (what bounds a sweep on the device). */
typedef struct sprs_hip_gauss_seidel_info {
    uint64_t iterations;
    double error;
    int32_t converged;
    uint64_t levels;
} sprs_hip_gauss_seidel_info;

/* Twin of gauss_seidel(mat, x, rhs, max_iter, eps) (heat.rs:103-139) with device-resident vectors:
 * up to max_iter sweeps  x[row] = (rhs[row] - sum_{col != row} val * x[col]) / diag  over the rows in
 * order, x updated IN PLACE (x_dev is the start vector and the result), each followed by the
 * reference's convergence test  error = sqrt(sum_i ((A x)_i - rhs_i)) < eps  (the SIGNED sum of the
 * residual, as the reference has it: a negative sum gives NaN and the sweeps go on).  A sweep is a
 * recurrence over the rows; on the device the rows run in dependency-level order (computed once per
 * handle) and a row reads this sweep's value of an earlier row as soon as that row has published
 * it, so every x[row] is computed from the same operands in the same order as on the CPU: the
 * iterates are bit-identical to the reference's.  `error` is A x by sprs_hip_spmv_f64 and a fixed
 * tree sum (ndarray's eight-accumulator sum rounds differently: equal to ~1e-13 of sum |r_i|).
 * A: square CSR handle with fewer than 2^32 rows (SPRS_HIP_STORAGE_MISMATCH for CSC: the reference's
 * outer_iterator would sweep the transpose); x_dev, rhs_dev: n doubles, not aliased.
 * SPRS_HIP_DIM_MISMATCH unless A.rows == A.cols == n (heat.rs:109-110); SPRS_HIP_BAD_STRUCTURE for a
 * row without a stored diagonal entry when a sweep would reach it (`diag.unwrap()`, heat.rs:127).
 * Running out of sweeps is not an error (Err(error) in the reference): status OK, converged = 0.
 * Blocks until done (the convergence test needs `error` on the host).  Deterministic. */
int32_t sprs_hip_gauss_seidel_f64(sprs_hip_csmat *a, double *x_dev, const double *rhs_dev, uint64_t n,
                                  uint64_t max_iter, double eps, sprs_hip_gauss_seidel_info *info,
                                  void *stream);

/* ---- SpGEMM ------------------------------------------------------------- */

/* Twin of smmp::mul_csr_csr (smmp.rs:196-416): C = A * B, all CSR, same index
 * types as the operands (smmp.rs:196-199).  Returns a NEW owning handle:
 * shape (A.rows, B.cols), zero-based indptr, every row strictly increasing
 * (sort_unstable, smmp.rs:126), structural zeros kept (no value test,
 * smmp.rs:109-119).  SPRS_HIP_DIM_MISMATCH unless A.cols == B.rows;
 * SPRS_HIP_STORAGE_MISMATCH unless both are CSR and share index widths;
 * SPRS_HIP_INDEX_OVERFLOW if nnz(C) does not fit Iptr.  Synchronous. */
int32_t sprs_hip_spgemm_f64(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c);

/* The two halves of smmp::mul_csr_csr as the reference exposes them:
 * sprs_hip_spgemm_symbolic — twin of smmp::symbolic (smmp.rs:81-131): the STRUCTURE of C = A * B (indptr and
 *   the sorted indices of every row, structural zeros kept) as a new handle whose values are 0.0;
 * sprs_hip_spgemm_numeric  — twin of smmp::numeric (smmp.rs:151-189): the VALUES of the product into `c`,
 *   which must have the product's structure (what _symbolic returned, for these or any operands with the same
 *   patterns — the re-use the reference's split exists for).  Unlike the reference, a `c` with another
 *   structure is detected: SPRS_HIP_BAD_STRUCTURE (nnz or indptr differ).  Same contract checks and
 *   status codes as sprs_hip_spgemm_f64; c's shape / storage / index types are checked like numeric()'s asserts
 *   (smmp.rs:161-166).  Both block until done. */
int32_t sprs_hip_spgemm_symbolic(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c_structure);
int32_t sprs_hip_spgemm_numeric(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c);

/* The same split with the symbolic work KEPT, as the reference's callers keep (indptr, indices) between symbolic and
 * numeric (smmp.rs:81-131 -> 151-189; sprs-benches/src/main.rs:211-260 re-multiplies the same operands):
 *   sprs_hip_spgemm_plan_create     the symbolic phase for (a, b): per-row product counts, how the rows are cut into
 *                                   tasks, the exact output counts, their prefix sum = C.indptr, nnz(C)
 *   sprs_hip_spgemm_plan_nnz        nnz of the product
 *   sprs_hip_spgemm_plan_structure  C with indptr + sorted indices, values 0.0            (what smmp::symbolic returns)
 *   sprs_hip_spgemm_plan_product    C complete: structure and values in one numeric pass  (== sprs_hip_spgemm_f64)
 *   sprs_hip_spgemm_plan_numeric    the VALUES into a c of the product's structure (shape, nnz, indptr checked ->
 *                                   SPRS_HIP_BAD_STRUCTURE; c's indices are not touched): only the value kernels run
 * a and b must be the handles (same structure buffers) the plan was made for, else SPRS_HIP_INVALID_ARG; their VALUES
 * may have changed in place in between.  All of these block until done. */
typedef struct sprs_hip_spgemm_plan sprs_hip_spgemm_plan;
int32_t sprs_hip_spgemm_plan_create(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_spgemm_plan **plan);
int32_t sprs_hip_spgemm_plan_nnz(const sprs_hip_spgemm_plan *plan, uint64_t *nnz);
int32_t sprs_hip_spgemm_plan_structure(sprs_hip_spgemm_plan *plan, const sprs_hip_csmat *a, const sprs_hip_csmat *b,
                                       sprs_hip_csmat **c_structure);
int32_t sprs_hip_spgemm_plan_product(sprs_hip_spgemm_plan *plan, const sprs_hip_csmat *a, const sprs_hip_csmat *b,
                                     sprs_hip_csmat **c);
int32_t sprs_hip_spgemm_plan_numeric(sprs_hip_spgemm_plan *plan, const sprs_hip_csmat *a, const sprs_hip_csmat *b,
                                     sprs_hip_csmat *c);
int32_t sprs_hip_spgemm_plan_free(sprs_hip_spgemm_plan *plan);

/* Storage conversion raw::convert_mat_storage / to_other_storage
 * (csmat.rs:1405-1426, 1782-1829): new owning handle with the other storage
 * order.  SPRS_HIP_INDEX_OVERFLOW where the reference panics (csmat.rs:1794). */
int32_t sprs_hip_csmat_to_other_storage(const sprs_hip_csmat *m, sprs_hip_csmat **out);

/* `&lhs * &rhs` for two sparse matrices: csmat_mul_csmat (csmat.rs:1895-1949), the storage dispatch around
 * smmp::mul_csr_csr — (CSR,CSR) multiplies directly, (CSR,CSC) converts the rhs, (CSC,*) multiplies the transpose
 * views the other way round and transposes back; the result has the storage of the lhs.  This is what the `Mul` impls
 * of the host mirrors call.  Status codes as sprs_hip_spgemm_f64 and sprs_hip_csmat_to_other_storage. */
int32_t sprs_hip_csmat_mul_csmat(const sprs_hip_csmat *lhs, const sprs_hip_csmat *rhs, sprs_hip_csmat **out);

/* ---- row-sharded SpMV over the GPUs of one node (one process per GPU, RCCL over xGMI) ----
 * The reference has no distributed code; the shard is its slice_outer (slicing.rs:65-89) with a rebased indptr
 * (indptr.rs:206-214): rank g owns rows [row_starts[g], row_starts[g+1]) and a full replica of x; one exchange, an
 * all-gather-v of y, issued as ONE group of direct sends / receives with every peer (xGMI is a point-to-point mesh).
 *   sprs_hip_dist_unique_id  128 bytes from ncclGetUniqueId: call on ONE rank and hand them to the others (MPI, a
 *                            torch.distributed broadcast, a file ...), like every NCCL / RCCL program does
 *   sprs_hip_dist_create     collective.  local_block = this rank's rows as a CSR handle with all `cols` columns and a
 *                            zero-based indptr (sprs_hip_csmat_slice_outer makes one).  It is cut into nsub sub-blocks of
 *                            equal cost; the exchange of a finished sub-block overlaps the multiply of the next
 *                            (nsub = 1: multiply, then exchange).  Every rank passes the same world, row_starts, nsub.
 *   sprs_hip_dist_spmv_f64   collective.  y (length rows, on this rank's device) = A * x (x: length cols, replicated).
 *                            Asynchronous on `stream`; y is complete for work queued on `stream` afterwards.
 * RCCL is loaded at run time (dlopen); world = 1 needs none.  SPRS_HIP_HIP_ERROR carries RCCL's message.
 * id_128_bytes = NULL with world > 1: no communicator is made, the handle exchanges over the peer route only (below). */
typedef struct sprs_hip_dist sprs_hip_dist;
int32_t sprs_hip_dist_unique_id(void *id_128_bytes);
int32_t sprs_hip_dist_create(sprs_hip_dist **d, const void *id_128_bytes, int32_t world, int32_t rank, uint64_t rows,
                             uint64_t cols, const uint64_t *row_starts /* world + 1 */, const sprs_hip_csmat *local_block,
                             int32_t nsub);
int32_t sprs_hip_dist_spmv_f64(sprs_hip_dist *d, const double *x_dev, uint64_t x_len, double *y_dev, uint64_t y_len,
                               void *stream);
/* the rank count the RCCL communicator itself reports (ncclCommCount); a world of one has no communicator and reports 1 */
int32_t sprs_hip_dist_comm_count(const sprs_hip_dist *d, int32_t *ranks);
/* THE SECOND EXCHANGE ROUTE: stores into the peers' receive windows over xGMI instead of ncclSend / ncclRecv (both
 * routes, to be compared by the first multi-GPU run).  Every rank exports its window (sprs_hip_dist_peer_handle: 64 bytes, a
 * HIP IPC handle), the host program hands all of them to every rank in rank order — like the RCCL id — and
 * sprs_hip_dist_peer_connect maps them; sprs_hip_dist_set_route(d, SPRS_HIP_ROUTE_PEER) then makes sprs_hip_dist_spmv_f64
 * push every finished sub-block into all peers' windows with one kernel (all links at once), publish an epoch word, wait for
 * the peers' epoch words and copy their rows into y.  Same result as the RCCL route, bit for bit (the multiply is the same
 * kernel).  A handle made with id_128_bytes = NULL and world > 1 has no RCCL communicator and can ONLY use this route (ranks
 * that share a device; hosts without RCCL).  All three calls are collective in the sense that every rank must make them; a
 * peer that does not arrive within 2 s turns the NEXT call into SPRS_HIP_HIP_ERROR instead of hanging the device.
 * sprs_hip_dist_free must not run while a peer may still store into this rank's window (synchronise the ranks first). */
#define SPRS_HIP_ROUTE_RCCL 0
#define SPRS_HIP_ROUTE_PEER 1
int32_t sprs_hip_dist_peer_handle(sprs_hip_dist *d, void *handle_64_bytes);
int32_t sprs_hip_dist_peer_connect(sprs_hip_dist *d, const void *handles /* world x 64 bytes, rank order */, int32_t world);
int32_t sprs_hip_dist_set_route(sprs_hip_dist *d, int32_t route);
int32_t sprs_hip_dist_route(const sprs_hip_dist *d, int32_t *route);
int32_t sprs_hip_dist_free(sprs_hip_dist *d);

/* Triplet (COO) assembly: twin of TriMatBase::to_csr / to_csc (triplet.rs:262-276) = TriMatIter::into_cs
 * (triplet_iter.rs:127-224): the n triplets (row_inds[p], col_inds[p], data[p]) — arrays in DEVICE memory, indices of
 * in_idx_bytes (4 or 8) each — are sorted by (outer, inner) with a stable device radix sort, duplicates are summed in
 * triplet order (`slot = slot + next`, triplet_iter.rs:168-171; the reference's unstable sort leaves that order
 * open), explicit zeros stay stored, empty outer slices get their indptr entries.  New owning handle with `storage`,
 * indices of out_idx_bytes and indptr of out_iptr_bytes.  SPRS_HIP_INVALID_ARG for an index out of bounds (add_triplet
 * asserts it, triplet.rs:171-172) or more than 2^32 rows / columns; SPRS_HIP_INDEX_OVERFLOW as CsMat construction. */
int32_t sprs_hip_triplets_to_cs(uint64_t rows, uint64_t cols, uint64_t n, const void *row_inds_dev, const void *col_inds_dev,
                                int32_t in_idx_bytes, const double *data_dev, int32_t storage, int32_t out_idx_bytes,
                                int32_t out_iptr_bytes, sprs_hip_csmat **out);

/* ---- tuning knobs (A/B runs; never needed for correctness) -------------- */

/* name = "spmv_kernel":    0 auto, 1 nnz-tiled streaming kernel, 2 wave-per-row (A/B only);
 * name = "spmv_xcs":       XCD-sliced plan for long rows: 0 auto, 1 force on, 2 off;
 * name = "spmv_xcs_split": rows with at least this many entries go to the sliced part (default 32);
 * name = "spmv_xcs_idx32": 1 (default): the sliced plan's own copies hold 32-bit column ids when cols < 2^32;
 * name = "spmv_sort_tiles": 1: tiles of the sliced plan's copies are stored sorted by column (default 0: measured slower);
 * name = "spmv_tile":      nnz per workgroup tile: 0 auto (default), 2048 or 4096;
 * name = "spmv_band*":     the banded plan (hot columns from LDS): spmv_band 0 auto / 1 on / 2 off, spmv_band_hot (slices, default
 *                          128), spmv_band_tile (labels per slice: 16384 or 8192), spmv_band_split (row length from which a row
 *                          is cut into pieces, default 24), spmv_band_rounds, spmv_band_hot_run, spmv_band_cold_tiles,
 *                          spmv_band_overlap, spmv_band_split_permute, spmv_band_phases, spmv_band_hot_cut;
 * name = "spgemm_ordered":  1 (default): SpGEMM adds the products of an entry in the reference's order (values bit-identical
 *                          to sprs', deterministic); 0: unordered atomic adds in the large-row kernel (same products, rounding-
 *                          level differences, not reproducible run to run; since round 4 no faster: the order costs nothing
 *                          once a wave instruction's products go out as ONE ds_add_f64, option spgemm_lane_order);
 * name = "gauss_seidel_blocks": workgroups of the Gauss-Seidel sweep kernel (0 = default: one per CU), "gauss_seidel_naps";
 * name = "spgemm_*", "spmm_long_row", "spmm_stream", "pool", "pool_max_bytes": INTEGRATION.md, "Options".
 * The whole table (name, default, range) is SPRS_HIP_OPTIONS in sprs_amd/csrc/common.hpp.  Developer switches — timing
 * experiments with WRONG results (spgemm_debug, spmv_xmask, spmv_band_debug) and the profiling printout spgemm_prof — exist
 * only in libraries built with -DSPRS_HIP_DEVTOOLS; this one rejects them.  get_option("devtools") tells which build it is.
 * Process-wide.  Unknown names / bad values return SPRS_HIP_INVALID_ARG. */
/* Contract: every option has a closed range (the table in sprs_amd/csrc/common.hpp); a value outside it — including a value
 * other than 0 / 1 for an on / off switch — and an unknown or retired name return SPRS_HIP_INVALID_ARG and change nothing
 * (sprs_hip_last_error() names the option and its range).  Nothing is coerced. */
int32_t sprs_hip_set_option(const char *name, int64_t value);
int32_t sprs_hip_get_option(const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* SPRS_HIP_H */
