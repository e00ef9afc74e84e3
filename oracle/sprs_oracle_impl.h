/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement of the sprs (0.11.5) SpMV / SpGEMM hot path, instantiated
 * once per (index type, indptr type) pair by sprs_oracle.c.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function below
 * against the reference's own fixtures and known-answer tests
 * (sprs/src/test_data.rs:6-124, sprs/src/sparse/prod.rs:375-458,
 * sprs/src/sparse/smmp.rs:422-513, sprs/tests/block_matrix.rs:71-108).
 *
 * The reference is Rust; rustc/cargo are absent in this environment, so this
 * is a restatement in C of the algorithms (never of the source text).  All
 * file:line citations are relative to /root/reference/.
 *
 * Build flags that matter: -ffp-contract=off, because sprs' MulAcc is
 * `*self += a * b` — a separately rounded multiply, then add
 * (sprs/src/mul_acc.rs:23-31).
 *
 * Macros expected: IDX_T, PTR_T, SUF(name).
 */

/* ---- SpMV ------------------------------------------------------------- */

/* prod::mul_acc_mat_vec_csr  (sprs/src/sparse/prod.rs:103-127).
 * y[i] += sum_p data[p] * x[indices[p]], p ascending, starting from the value
 * already in y[i].  indptr may be "non-proper" (not zero based), as produced
 * by slice_outer (sprs/src/sparse/indptr.rs:216-219, 252-274): `indices` and
 * `data` then point at the element addressed by indptr[0].
 * Returns ORACLE_DIM_MISMATCH where the reference panics "Dimension mismatch"
 * (prod.rs:114-117). */
int SUF(oracle_mul_acc_mat_vec_csr)(uint64_t rows, uint64_t cols,
                                    const PTR_T *indptr, const IDX_T *indices,
                                    const double *data, const double *x,
                                    uint64_t x_len, double *y, uint64_t y_len)
{
    if (cols != x_len || rows != y_len)
        return ORACLE_DIM_MISMATCH;
    const uint64_t off = rows ? (uint64_t)indptr[0] : 0;
    for (uint64_t i = 0; i < rows; ++i) {          /* outer_iterator, prod.rs:120 */
        double tv = y[i];                          /* res_vec.index_mut(row_ind)   */
        const uint64_t e = (uint64_t)indptr[i + 1] - off;
        for (uint64_t p = (uint64_t)indptr[i] - off; p < e; ++p) {
            const double prod = data[p] * x[indices[p]];   /* a * b   (mul_acc.rs:29) */
            tv += prod;                                    /* += ...  (mul_acc.rs:29) */
        }
        y[i] = tv;
    }
    return ORACLE_OK;
}

/* NOT IN THE REFERENCE (sprs SpMV is single threaded): the same
 * loop with the rows split over OpenMP threads.  Present only because the
 * north star asks for an all-host-cores CPU figure; per-row arithmetic and
 * order are unchanged, so results are bit-identical to the serial form. */
int SUF(oracle_mul_acc_mat_vec_csr_omp)(uint64_t rows, uint64_t cols,
                                        const PTR_T *indptr, const IDX_T *indices,
                                        const double *data, const double *x,
                                        uint64_t x_len, double *y, uint64_t y_len,
                                        int nthreads)
{
    if (cols != x_len || rows != y_len)
        return ORACLE_DIM_MISMATCH;
    const uint64_t off = rows ? (uint64_t)indptr[0] : 0;
    if (nthreads < 1) nthreads = 1;
    /* contiguous row blocks of ~equal nnz, so hubs do not serialise a thread */
    const uint64_t nnz = rows ? (uint64_t)indptr[rows] - off : 0;
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        uint64_t lo_nnz = nnz / (uint64_t)nt * (uint64_t)t;
        uint64_t hi_nnz = (t + 1 == nt) ? nnz : nnz / (uint64_t)nt * (uint64_t)(t + 1);
        /* first row whose start >= lo_nnz */
        uint64_t a = 0, b = rows;
        while (a < b) { uint64_t m = (a + b) / 2; if ((uint64_t)indptr[m] - off < lo_nnz) a = m + 1; else b = m; }
        uint64_t r0 = (t == 0) ? 0 : a;
        a = 0; b = rows;
        while (a < b) { uint64_t m = (a + b) / 2; if ((uint64_t)indptr[m] - off < hi_nnz) a = m + 1; else b = m; }
        uint64_t r1 = (t + 1 == nt) ? rows : a;
        for (uint64_t i = r0; i < r1; ++i) {
            double tv = y[i];
            const uint64_t e = (uint64_t)indptr[i + 1] - off;
            for (uint64_t p = (uint64_t)indptr[i] - off; p < e; ++p) {
                const double prod = data[p] * x[indices[p]];
                tv += prod;
            }
            y[i] = tv;
        }
    }
    return ORACLE_OK;
}

/* ---- SpGEMM: SMMP ----------------------------------------------------- */

typedef struct {
    IDX_T *ptr;
    uint64_t len, cap;
} SUF(idxvec);

static void SUF(idxvec_reserve)(SUF(idxvec) *v, uint64_t cap)
{
    if (cap <= v->cap) return;
    IDX_T *p = (IDX_T *)realloc(v->ptr, (size_t)(cap ? cap : 1) * sizeof(IDX_T));
    if (!p) { fprintf(stderr, "oracle: out of memory (%llu indices)\n", (unsigned long long)cap); abort(); }
    v->ptr = p;
    v->cap = cap;
}

static inline void SUF(idxvec_push)(SUF(idxvec) *v, IDX_T x)
{
    if (v->len == v->cap)
        SUF(idxvec_reserve)(v, v->cap < 16 ? 16 : v->cap * 2);   /* Vec growth */
    v->ptr[v->len++] = x;
}

/* sort_unstable of one output row (smmp.rs:126).  Keys are unique, so every
 * correct sort yields the same array; this is a median-of-3 quicksort with an
 * insertion-sort floor. */
static void SUF(sort_idx)(IDX_T *a, int64_t n)
{
    while (n > 24) {
        IDX_T x = a[0], y = a[n / 2], z = a[n - 1], piv;
        if (x < y) piv = (y < z) ? y : (x < z ? z : x);
        else       piv = (x < z) ? x : (y < z ? z : y);
        int64_t i = 0, j = n - 1;
        for (;;) {
            while (a[i] < piv) ++i;
            while (a[j] > piv) --j;
            if (i >= j) break;
            IDX_T t = a[i]; a[i] = a[j]; a[j] = t;
            ++i; --j;
        }
        /* recurse on the smaller side, loop on the larger */
        int64_t left = j + 1, right = n - left;
        if (left < right) { SUF(sort_idx)(a, left); a += left; n = right; }
        else              { SUF(sort_idx)(a + left, right); n = left; }
    }
    for (int64_t i = 1; i < n; ++i) {
        IDX_T k = a[i];
        int64_t j = i - 1;
        while (j >= 0 && a[j] > k) { a[j + 1] = a[j]; --j; }
        a[j + 1] = k;
    }
}

/* smmp::symbolic  (sprs/src/sparse/smmp.rs:81-131).
 * a_* describe a (possibly sliced, non-proper indptr) chunk of A's rows;
 * c_indptr (a_rows+1 entries) comes out zero based (smmp.rs:101);
 * seen has b_cols entries and is all-false on exit (smmp.rs:127-129).
 * Returns ORACLE_INDEX_OVERFLOW where I::from_usize / Iptr::from_usize would
 * panic (smmp.rs:116,121; sprs/src/indexing.rs:104-108). */
static int SUF(symbolic)(uint64_t a_rows, const PTR_T *a_indptr, const IDX_T *a_indices,
                         const PTR_T *b_indptr, const IDX_T *b_indices, uint64_t b_cols,
                         PTR_T *c_indptr, SUF(idxvec) *c_indices, uint8_t *seen,
                         uint64_t a_nnz, uint64_t b_nnz)
{
    c_indices->len = 0;                                          /* clear()          */
    SUF(idxvec_reserve)(c_indices, a_nnz + b_nnz);               /* reserve_exact    */
    memset(seen, 0, (size_t)b_cols);                             /* smmp.rs:97-99    */
    const uint64_t a_off = a_rows ? (uint64_t)a_indptr[0] : 0;
    const uint64_t b_off = (uint64_t)b_indptr[0];
    c_indptr[0] = 0;
    for (uint64_t a_row = 0; a_row < a_rows; ++a_row) {          /* iter_outer_sz    */
        uint64_t length = 0;
        const uint64_t ae = (uint64_t)a_indptr[a_row + 1] - a_off;
        for (uint64_t ap = (uint64_t)a_indptr[a_row] - a_off; ap < ae; ++ap) {
            const uint64_t b_row = a_indices[ap];
            const uint64_t be = (uint64_t)b_indptr[b_row + 1] - b_off;
            for (uint64_t bp = (uint64_t)b_indptr[b_row] - b_off; bp < be; ++bp) {
                const uint64_t b_col = b_indices[bp];
                if (!seen[b_col]) {                              /* no value test: structural zeros kept */
                    seen[b_col] = 1;
                    SUF(idxvec_push)(c_indices, (IDX_T)b_col);
                    ++length;
                }
            }
        }
        const uint64_t c_start = (uint64_t)c_indptr[a_row];
        const uint64_t c_end = c_start + length;
        if ((uint64_t)(PTR_T)c_end != c_end)
            return ORACLE_INDEX_OVERFLOW;
        c_indptr[a_row + 1] = (PTR_T)c_end;
        SUF(sort_idx)(c_indices->ptr + c_start, (int64_t)length); /* sort_unstable    */
        for (uint64_t p = c_start; p < c_end; ++p)
            seen[c_indices->ptr[p]] = 0;
    }
    return ORACLE_OK;
}

/* smmp::numeric  (sprs/src/sparse/smmp.rs:151-189).
 * tmp has b_cols entries; zeroed on entry (smmp.rs:169-171), zero on exit
 * (swap-to-zero, smmp.rs:183-187).  c_indptr may be non-proper; c_indices /
 * c_data point at the element addressed by c_indptr[0]. */
static void SUF(numeric)(uint64_t a_rows, const PTR_T *a_indptr, const IDX_T *a_indices,
                         const double *a_data, const PTR_T *b_indptr, const IDX_T *b_indices,
                         const double *b_data, uint64_t b_cols, const PTR_T *c_indptr,
                         const IDX_T *c_indices, double *c_data, double *tmp)
{
    for (uint64_t i = 0; i < b_cols; ++i) tmp[i] = 0.0;
    const uint64_t a_off = a_rows ? (uint64_t)a_indptr[0] : 0;
    const uint64_t b_off = (uint64_t)b_indptr[0];
    const uint64_t c_off = a_rows ? (uint64_t)c_indptr[0] : 0;
    for (uint64_t row = 0; row < a_rows; ++row) {
        const uint64_t ae = (uint64_t)a_indptr[row + 1] - a_off;
        for (uint64_t ap = (uint64_t)a_indptr[row] - a_off; ap < ae; ++ap) {
            const uint64_t b_row = a_indices[ap];
            const double a_val = a_data[ap];
            const uint64_t be = (uint64_t)b_indptr[b_row + 1] - b_off;
            for (uint64_t bp = (uint64_t)b_indptr[b_row] - b_off; bp < be; ++bp) {
                const double prod = a_val * b_data[bp];
                tmp[b_indices[bp]] += prod;                      /* mul_acc, unfused */
            }
        }
        const uint64_t ce = (uint64_t)c_indptr[row + 1] - c_off;
        for (uint64_t cp = (uint64_t)c_indptr[row] - c_off; cp < ce; ++cp) {
            const uint64_t c_col = c_indices[cp];
            c_data[cp] = tmp[c_col];                             /* mem::swap with zero */
            tmp[c_col] = 0.0;
        }
    }
}

/* Public, single-chunk forms of the two passes, as the reference's own
 * `symbolic_and_numeric` test drives them (smmp.rs:422-465).  c_indices is
 * returned in a malloc'd buffer (free with oracle_free). */
int SUF(oracle_symbolic)(uint64_t a_rows, uint64_t a_cols, const PTR_T *a_indptr,
                         const IDX_T *a_indices, uint64_t b_rows, uint64_t b_cols,
                         const PTR_T *b_indptr, const IDX_T *b_indices, PTR_T *c_indptr,
                         IDX_T **c_indices_out, uint64_t *c_nnz_out)
{
    if (a_cols != b_rows) return ORACLE_DIM_MISMATCH;            /* smmp.rs:95 */
    if ((uint64_t)(IDX_T)(b_cols ? b_cols - 1 : 0) != (b_cols ? b_cols - 1 : 0))
        return ORACLE_INDEX_OVERFLOW;
    SUF(idxvec) v = {0, 0, 0};
    uint8_t *seen = (uint8_t *)malloc((size_t)(b_cols ? b_cols : 1));
    const uint64_t a_nnz = a_rows ? (uint64_t)a_indptr[a_rows] - (uint64_t)a_indptr[0] : 0;
    const uint64_t b_nnz = (uint64_t)b_indptr[b_rows] - (uint64_t)b_indptr[0];
    int st = SUF(symbolic)(a_rows, a_indptr, a_indices, b_indptr, b_indices, b_cols,
                           c_indptr, &v, seen, a_nnz, b_nnz);
    free(seen);
    if (st != ORACLE_OK) { free(v.ptr); return st; }
    *c_indices_out = v.ptr;
    *c_nnz_out = v.len;
    return ORACLE_OK;
}

int SUF(oracle_numeric)(uint64_t a_rows, uint64_t a_cols, const PTR_T *a_indptr,
                        const IDX_T *a_indices, const double *a_data, uint64_t b_rows,
                        uint64_t b_cols, const PTR_T *b_indptr, const IDX_T *b_indices,
                        const double *b_data, const PTR_T *c_indptr, const IDX_T *c_indices,
                        double *c_data)
{
    if (a_cols != b_rows) return ORACLE_DIM_MISMATCH;            /* smmp.rs:163 */
    double *tmp = (double *)malloc((size_t)(b_cols ? b_cols : 1) * sizeof(double));
    SUF(numeric)(a_rows, a_indptr, a_indices, a_data, b_indptr, b_indices, b_data, b_cols,
                 c_indptr, c_indices, c_data, tmp);
    free(tmp);
    return ORACLE_OK;
}

/* smmp::mul_csr_csr + mul_csr_csr_with_workspace  (smmp.rs:196-237, 256-416).
 *
 * threads  > 0 : ThreadingStrategy::Fixed(threads)
 * threads == 0 : ThreadingStrategy::Automatic, with num_cpus = omp_get_num_procs()
 *
 * Both of the reference's chunkings are kept: equal ROW chunks for the
 * symbolic pass (smmp.rs:277-296) and ~equal nnz(C) chunks for the numeric
 * pass (smmp.rs:332-372), with the serial concatenation + prefix sum between
 * them (smmp.rs:320-331).  rayon's par_iter is rendered as one OpenMP thread
 * per chunk.  Output buffers are malloc'd (free with oracle_free); the result
 * does not depend on `threads` (sprs-benches/src/main.rs:233,246,259). */
int SUF(oracle_mul_csr_csr)(uint64_t a_rows, uint64_t a_cols, const PTR_T *a_indptr,
                            const IDX_T *a_indices, const double *a_data, uint64_t b_rows,
                            uint64_t b_cols, const PTR_T *b_indptr, const IDX_T *b_indices,
                            const double *b_data, int threads, PTR_T **c_indptr_out,
                            IDX_T **c_indices_out, double **c_data_out, uint64_t *c_nnz_out,
                            int *threads_used)
{
    if (a_cols != b_rows) return ORACLE_DIM_MISMATCH;            /* smmp.rs:207 */
    if (b_cols && (uint64_t)(IDX_T)(b_cols - 1) != b_cols - 1) return ORACLE_INDEX_OVERFLOW;
    const uint64_t a_off = a_rows ? (uint64_t)a_indptr[0] : 0;
    const uint64_t a_nnz = a_rows ? (uint64_t)a_indptr[a_rows] - a_off : 0;
    const uint64_t b_nnz = (uint64_t)b_indptr[b_rows] - (uint64_t)b_indptr[0];

    /* thread-count rule, smmp.rs:210-227 */
    uint64_t want;
    if (threads > 0) {
        want = (uint64_t)threads;
    } else {
        const uint64_t nb_cpus = (uint64_t)omp_get_num_procs();
        const uint64_t ideal_chunk_size = 8128;
        uint64_t wanted_threads = (a_nnz + b_nnz) / ideal_chunk_size;
        if (wanted_threads < 1) wanted_threads = 1;
        want = wanted_threads < nb_cpus ? wanted_threads : nb_cpus;
    }
    const uint64_t rows_or_1 = a_rows > 1 ? a_rows : 1;
    const uint64_t nb = want < rows_or_1 ? want : rows_or_1;
    if (threads_used) *threads_used = (int)nb;
    const uint64_t workspace_len = b_cols;

    /* ---- symbolic over equal-row chunks (smmp.rs:277-319) ---- */
    const uint64_t chunk_size = (a_rows + 1) / nb;               /* lhs.indptr().len() / nb_threads */
    uint64_t *starts = (uint64_t *)malloc(sizeof(uint64_t) * (nb + 1));
    PTR_T **ip_chunks = (PTR_T **)calloc(nb, sizeof(PTR_T *));
    SUF(idxvec) *ix_chunks = (SUF(idxvec) *)calloc(nb, sizeof(SUF(idxvec)));
    int *status = (int *)calloc(nb, sizeof(int));
    for (uint64_t c = 0; c < nb; ++c) {
        starts[c] = c * chunk_size;
        uint64_t stop = (c + 1 < nb) ? (c + 1) * chunk_size : a_rows;
        starts[c + 1] = stop;
        ip_chunks[c] = (PTR_T *)calloc(stop - starts[c] + 1, sizeof(PTR_T));
    }
#pragma omp parallel for schedule(static, 1) num_threads((int)nb)
    for (int64_t c = 0; c < (int64_t)nb; ++c) {
        uint8_t *seen = (uint8_t *)malloc((size_t)(workspace_len ? workspace_len : 1));
        const uint64_t r0 = starts[c], r1 = starts[c + 1];
        /* lhs.slice_outer(start..stop): indptr not rebased, indices offset
         * (sprs/src/sparse/slicing.rs:65-89) */
        const PTR_T *ip = a_indptr + r0;
        const uint64_t first = (r1 > r0) ? (uint64_t)ip[0] - a_off : 0;
        const uint64_t cnnz = (r1 > r0) ? (uint64_t)ip[r1 - r0] - (uint64_t)ip[0] : 0;
        status[c] = SUF(symbolic)(r1 - r0, ip, a_indices + first, b_indptr, b_indices, b_cols,
                                  ip_chunks[c], &ix_chunks[c], seen, cnnz, b_nnz);
        free(seen);
    }
    int st = ORACLE_OK;
    for (uint64_t c = 0; c < nb; ++c) if (status[c] != ORACLE_OK) st = status[c];

    /* ---- concatenate + serial prefix sum (smmp.rs:320-331) ---- */
    uint64_t c_nnz = 0;
    for (uint64_t c = 0; c < nb; ++c) c_nnz += ix_chunks[c].len;
    IDX_T *res_indices = (IDX_T *)malloc((size_t)(c_nnz ? c_nnz : 1) * sizeof(IDX_T));
    PTR_T *res_indptr = (PTR_T *)malloc((size_t)(a_rows + 1) * sizeof(PTR_T));
    if (st == ORACLE_OK) {
        uint64_t w = 0;
        for (uint64_t c = 0; c < nb; ++c) {
            memcpy(res_indices + w, ix_chunks[c].ptr, (size_t)ix_chunks[c].len * sizeof(IDX_T));
            w += ix_chunks[c].len;
        }
        uint64_t last = 0, row = 0;
        res_indptr[0] = 0;
        for (uint64_t c = 0; c < nb && st == ORACLE_OK; ++c) {
            const uint64_t n = starts[c + 1] - starts[c];
            for (uint64_t i = 0; i < n; ++i) {
                last += (uint64_t)ip_chunks[c][i + 1] - (uint64_t)ip_chunks[c][i];
                if ((uint64_t)(PTR_T)last != last) { st = ORACLE_INDEX_OVERFLOW; break; }
                res_indptr[++row] = (PTR_T)last;
            }
        }
    }
    for (uint64_t c = 0; c < nb; ++c) { free(ip_chunks[c]); free(ix_chunks[c].ptr); }
    free(ip_chunks); free(ix_chunks); free(status); free(starts);
    if (st != ORACLE_OK) { free(res_indices); free(res_indptr); return st; }

    /* ---- numeric over ~equal-nnz(C) chunks (smmp.rs:332-404) ---- */
    double *res_data = (double *)calloc((size_t)(c_nnz ? c_nnz : 1), sizeof(double));
    const uint64_t nchunk_size = c_nnz / nb;
    uint64_t *split_rows = (uint64_t *)malloc(sizeof(uint64_t) * (nb + 2));
    uint64_t *split_ends = (uint64_t *)malloc(sizeof(uint64_t) * (nb + 2));
    uint64_t nsplit = 0, split_nnz = 0, split_row = 0;
    for (uint64_t row = 0; row <= a_rows; ++row) {               /* res_indptr.iter().enumerate() */
        const uint64_t nnz = (uint64_t)res_indptr[row];
        if (nnz - split_nnz > nchunk_size && row > 0) {
            split_rows[nsplit] = split_row;                      /* lhs.slice_outer(split_row..row-1) */
            split_ends[nsplit] = row - 1;
            ++nsplit;
            split_nnz = nnz;
            split_row = row - 1;
        }
    }
    split_rows[nsplit] = split_row;                              /* tail chunk, smmp.rs:368-372 */
    split_ends[nsplit] = a_rows;
    ++nsplit;
    const uint64_t nthr = nsplit < nb ? nsplit : nb;             /* zip() stops at the shorter side */
#pragma omp parallel for schedule(static, 1) num_threads((int)nthr)
    for (int64_t c = 0; c < (int64_t)nthr; ++c) {
        double *tmp = (double *)malloc((size_t)(workspace_len ? workspace_len : 1) * sizeof(double));
        const uint64_t r0 = split_rows[c], r1 = split_ends[c];
        const uint64_t a_first = (r1 > r0) ? (uint64_t)a_indptr[r0] - a_off : 0;
        const uint64_t c_first = (uint64_t)res_indptr[r0];
        SUF(numeric)(r1 - r0, a_indptr + r0, a_indices + a_first, a_data + a_first, b_indptr,
                     b_indices, b_data, b_cols, res_indptr + r0, res_indices + c_first,
                     res_data + c_first, tmp);
        free(tmp);
    }
    free(split_rows); free(split_ends);
    *c_indptr_out = res_indptr;
    *c_indices_out = res_indices;
    *c_data_out = res_data;
    *c_nnz_out = c_nnz;
    return ORACLE_OK;
}

/* ---- containers / generators ------------------------------------------ */

/* CsMatI::eye  (sprs/src/sparse/csmat.rs:416-426): CSR identity. */
void SUF(oracle_eye)(uint64_t dim, PTR_T *indptr, IDX_T *indices, double *data)
{
    for (uint64_t i = 0; i <= dim; ++i) indptr[i] = (PTR_T)i;
    for (uint64_t i = 0; i < dim; ++i) { indices[i] = (IDX_T)i; data[i] = 1.0; }
}

/* grid_laplacian  (sprs/examples/heat.rs:45-80).  Border vertices get a single
 * diagonal 1.0 (Dirichlet), interior ones the 5-point stencil in the order
 * (i-1,j) (i,j-1) (i,j) (i,j+1) (i+1,j) with values 1 1 -4 1 1.  The reference
 * flattens a vertex as `i * rows + j` (heat.rs:60); kept as is.
 * Buffers: indptr rows*cols+1, indices/data sized by oracle_grid_laplacian_nnz. */
uint64_t SUF(oracle_grid_laplacian)(uint64_t rows, uint64_t cols, PTR_T *indptr,
                                    IDX_T *indices, double *data)
{
    uint64_t cumsum = 0, v = 0;
    for (uint64_t i = 0; i < rows; ++i) {
        for (uint64_t j = 0; j < cols; ++j) {
            indptr[v++] = (PTR_T)cumsum;
            const int border = (i == 0 || i == rows - 1 || j == 0 || j == cols - 1);
            if (border) {
                indices[cumsum] = (IDX_T)(i * rows + j); data[cumsum++] = 1.0;
            } else {
                indices[cumsum] = (IDX_T)((i - 1) * rows + j); data[cumsum++] = 1.0;
                indices[cumsum] = (IDX_T)(i * rows + j - 1);   data[cumsum++] = 1.0;
                indices[cumsum] = (IDX_T)(i * rows + j);       data[cumsum++] = -4.0;
                indices[cumsum] = (IDX_T)(i * rows + j + 1);   data[cumsum++] = 1.0;
                indices[cumsum] = (IDX_T)((i + 1) * rows + j); data[cumsum++] = 1.0;
            }
        }
    }
    indptr[v] = (PTR_T)cumsum;
    return cumsum;
}

/* raw::convert_mat_storage  (sprs/src/sparse/csmat.rs:1782-1829): counting
 * sort CSR(outer x inner) -> CSC of the same matrix, i.e. the CSR arrays of
 * the transpose.  Used by the (CSR,CSC) / (CSC,*) SpGEMM dispatch cases
 * (csmat.rs:1933-1948).  Returns ORACLE_INDEX_OVERFLOW where the reference
 * panics "Index type is not large enough to hold ..." (csmat.rs:1794-1797). */
int SUF(oracle_convert_storage)(uint64_t outer, uint64_t inner, uint64_t mat_rows,
                                const PTR_T *indptr, const IDX_T *indices, const double *data,
                                PTR_T *o_indptr, IDX_T *o_indices, double *o_data)
{
    /* the reference tests `mat.rows()` whatever the storage order (csmat.rs:1794) */
    if ((uint64_t)(IDX_T)mat_rows != mat_rows) return ORACLE_INDEX_OVERFLOW;
    const uint64_t off = (uint64_t)indptr[0];
    const uint64_t nnz = (uint64_t)indptr[outer] - off;
    for (uint64_t i = 0; i <= inner; ++i) o_indptr[i] = 0;
    for (uint64_t p = 0; p < nnz; ++p) o_indptr[indices[p] + 1]++;          /* histogram  */
    for (uint64_t i = 0; i < inner; ++i) o_indptr[i + 1] += o_indptr[i];    /* cumsum     */
    PTR_T *next = (PTR_T *)malloc((size_t)(inner + 1) * sizeof(PTR_T));
    memcpy(next, o_indptr, (size_t)(inner + 1) * sizeof(PTR_T));
    for (uint64_t r = 0; r < outer; ++r) {
        const uint64_t e = (uint64_t)indptr[r + 1] - off;
        for (uint64_t p = (uint64_t)indptr[r] - off; p < e; ++p) {
            const uint64_t dst = (uint64_t)next[indices[p]]++;
            o_indices[dst] = (IDX_T)r;
            o_data[dst] = data[p];
        }
    }
    free(next);
    return ORACLE_OK;
}

/* utils::check_compressed_structure  (sprs/src/sparse.rs:300-358), zero-based
 * or offset indptr.  Returns ORACLE_OK or ORACLE_BAD_STRUCTURE. */
int SUF(oracle_check_structure)(uint64_t inner, uint64_t outer, const PTR_T *indptr,
                                const IDX_T *indices, uint64_t indices_len)
{
    const uint64_t off = (uint64_t)indptr[0];
    for (uint64_t i = 0; i < outer; ++i)
        if (indptr[i + 1] < indptr[i]) return ORACLE_BAD_STRUCTURE;          /* Unsorted   */
    if ((uint64_t)indptr[outer] - off != indices_len) return ORACLE_BAD_STRUCTURE;
    for (uint64_t r = 0; r < outer; ++r) {
        const uint64_t s = (uint64_t)indptr[r] - off, e = (uint64_t)indptr[r + 1] - off;
        for (uint64_t p = s; p < e; ++p) {
            if ((uint64_t)indices[p] >= inner) return ORACLE_BAD_STRUCTURE;  /* OutOfRange */
            if (p > s && indices[p] <= indices[p - 1]) return ORACLE_BAD_STRUCTURE;
        }
    }
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * BiCGSTAB — restatement of sprs/src/sparse/linalg/bicgstab.rs (solver for A x = b, unpreconditioned).
 *
 * The reference keeps every vector as a CsVec and multiplies with `&CsMat * &CsVec`
 * (csr_mul_csvec, prod.rs:162-184: one sparse dot per row, ascending column; a CSC matrix takes the
 * SpGEMM route, which also adds ascending k).  For vectors without structural zeros that is, entry for
 * entry and addition for addition, the dense arithmetic below: dots and norms are serial sums from 0
 * (vec.rs:846-880, 907-918), axpys are unfused (`x * alpha`, then add: bicgstab.rs:199-210).
 * `a` is CSR here; a CSC operand is converted by the caller (oracle_convert_storage).
 * new(): :117-143, solve(): :148-171, soft/hard restart: :175-192, step(): :194-229.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t iteration_count, soft_restart_count, hard_restart_count;
    double err, rho;
    int32_t converged;
} SUF(oracle_bicgstab_info);

static void SUF(bicg_spmv)(uint64_t n, const PTR_T *ip, const IDX_T *ix, const double *dt, const double *x, double *y)
{
    for (uint64_t i = 0; i < n; ++i) {
        double sum = 0.0;                                   /* dot_acc: Acc::zero(), then mul_acc ascending */
        for (uint64_t p = (uint64_t)ip[i]; p < (uint64_t)ip[i + 1]; ++p) {
            const double prod = dt[p] * x[ix[p]];
            sum = sum + prod;
        }
        y[i] = sum;
    }
}

static double SUF(bicg_dot)(uint64_t n, const double *a, const double *b)
{
    double sum = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        const double prod = a[i] * b[i];
        sum = sum + prod;
    }
    return sum;
}

int SUF(oracle_bicgstab)(uint64_t n, const PTR_T *ip, const IDX_T *ix, const double *dt, const double *x0,
                         const double *b, double tol, uint64_t max_iter, double soft_restart_threshold,
                         double *x, SUF(oracle_bicgstab_info) *info)
{
    double *r = (double *)malloc(8 * n * 7 + 64);
    if (!r) return ORACLE_BAD_STRUCTURE;
    double *rhat = r + n, *p = rhat + n, *v = p + n, *s = v + n, *t = s + n, *h = t + n;
    uint64_t it = 0, soft = 0, hard = 0;
    /* new() */
    SUF(bicg_spmv)(n, ip, ix, dt, x0, v);
    for (uint64_t i = 0; i < n; ++i) { r[i] = b[i] - v[i]; rhat[i] = r[i]; p[i] = r[i]; x[i] = x0[i]; }
    double err = sqrt(SUF(bicg_dot)(n, r, r));
    double rho = err * err;
    int converged = 0;
    for (uint64_t k = 0; k < max_iter && !converged; ++k) {
        /* step() */
        ++it;
        SUF(bicg_spmv)(n, ip, ix, dt, p, v);
        const double alpha = rho / SUF(bicg_dot)(n, rhat, v);
        for (uint64_t i = 0; i < n; ++i) { const double q = p[i] * alpha; h[i] = x[i] + q; }
        for (uint64_t i = 0; i < n; ++i) { const double q = v[i] * alpha; s[i] = r[i] - q; }
        SUF(bicg_spmv)(n, ip, ix, dt, s, t);
        const double omega = SUF(bicg_dot)(n, t, s) / SUF(bicg_dot)(n, t, t);
        for (uint64_t i = 0; i < n; ++i) { const double q = omega * s[i]; x[i] = h[i] + q; }
        for (uint64_t i = 0; i < n; ++i) { const double q = t[i] * omega; r[i] = s[i] - q; }
        err = sqrt(SUF(bicg_dot)(n, r, r));
        const double rho_prev = rho;
        rho = SUF(bicg_dot)(n, rhat, r);
        if (fabs(rho) / (err * err) < soft_restart_threshold) {
            ++soft;                                           /* soft_restart() */
            for (uint64_t i = 0; i < n; ++i) { rhat[i] = r[i]; p[i] = r[i]; }
            rho = err * err;
        } else {
            const double beta = (rho / rho_prev) * (alpha / omega);
            for (uint64_t i = 0; i < n; ++i) {
                const double q = v[i] * omega;
                const double d = p[i] - q;
                const double e = d * beta;
                p[i] = r[i] + e;
            }
        }
        if (err < tol) {
            /* hard_restart(): true residual, then soft restart without counting it */
            ++hard;
            SUF(bicg_spmv)(n, ip, ix, dt, x, v);
            for (uint64_t i = 0; i < n; ++i) r[i] = b[i] - v[i];
            err = sqrt(SUF(bicg_dot)(n, r, r));
            for (uint64_t i = 0; i < n; ++i) { rhat[i] = r[i]; p[i] = r[i]; }
            rho = err * err;
            if (err < tol) converged = 1;
        }
    }
    info->iteration_count = it;
    info->soft_restart_count = soft;
    info->hard_restart_count = hard;
    info->err = err;
    info->rho = rho;
    info->converged = converged;
    free(r);
    return ORACLE_OK;
}


/* ---- Gauss-Seidel sweep of the heat example (the other caller that loops on the SpMV) ----
 * gauss_seidel()  (sprs/examples/heat.rs:103-139), statement for statement:
 *   for every sweep: for every row in order: sigma = sum of val * x[col] over the stored entries with col != row, in entry
 *   order, reading x IN PLACE (columns before the row already hold this sweep's values); diag = the entry with col == row
 *   (`diag.unwrap()`: a row without one panics, heat.rs:127 -> ORACLE_BAD_STRUCTURE); x[row] = (rhs[row] - sigma) / diag;
 *   after the sweep  error = (&mat * &x - rhs).sum().sqrt()  — the SIGNED sum of the residual, as the reference has it
 *   (a negative sum gives NaN, and `NaN < eps` keeps iterating) — and Ok((it, error)) as soon as error < eps, Err(error)
 *   after max_iter sweeps.  (The error computed before the first sweep, heat.rs:111, is only returned when max_iter = 0.)
 * `&mat * &x` is csr_mulacc_dense_colmaj on a zeroed vector (csmat.rs:2119-2158, prod.rs:274-298): the SpMV hot path.
 * `.sum()` of a contiguous ndarray is numeric_util::unrolled_fold — THIRD-PARTY crate ndarray (sprs/Cargo.toml:24,
 * ">=0.15.0, <0.18"; not vendored under /root/reference): eight interleaved partial sums p0..p7 over blocks of eight,
 * combined as ((((0 + (p0+p4)) + (p1+p5)) + (p2+p6)) + (p3+p7)), then the < 8 trailing elements one by one.  Restated
 * from the published source of ndarray 0.15 / 0.16 (src/numeric_util.rs); PARITY UNPINNED for this one function: the
 * example asserts nothing and no Rust toolchain is here, so the tests compare `error` to 1e-10 of sum |r_i| only, and pin x by
 * an independent line-by-line Python restatement and a dense solve (tests/test_oracle_golden.py).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t iterations;   /* Ok((it, _)): index of the sweep after which error < eps; Err: max_iter */
    double error;
    int32_t converged;     /* Ok = 1, Err = 0 */
} SUF(oracle_gauss_seidel_info);

static double SUF(ndarray_sum)(const double *xs, uint64_t n)
{
    double acc = 0.0, p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0, p5 = 0.0, p6 = 0.0, p7 = 0.0;
    while (n >= 8) {
        p0 = p0 + xs[0]; p1 = p1 + xs[1]; p2 = p2 + xs[2]; p3 = p3 + xs[3];
        p4 = p4 + xs[4]; p5 = p5 + xs[5]; p6 = p6 + xs[6]; p7 = p7 + xs[7];
        xs += 8; n -= 8;
    }
    acc = acc + (p0 + p4);
    acc = acc + (p1 + p5);
    acc = acc + (p2 + p6);
    acc = acc + (p3 + p7);
    for (uint64_t i = 0; i < n; ++i) acc = acc + xs[i];
    return acc;
}

static double SUF(gs_error)(uint64_t n, const PTR_T *ip, const IDX_T *ix, const double *dt, const double *x,
                            const double *rhs, double *work)
{
    SUF(bicg_spmv)(n, ip, ix, dt, x, work);                    /* &mat * &x: 0 + products in entry order */
    for (uint64_t i = 0; i < n; ++i) work[i] = work[i] - rhs[i];
    return sqrt(SUF(ndarray_sum)(work, n));
}

int SUF(oracle_gauss_seidel)(uint64_t n, const PTR_T *ip, const IDX_T *ix, const double *dt, double *x,
                             const double *rhs, uint64_t max_iter, double eps, SUF(oracle_gauss_seidel_info) *info)
{
    double *work = (double *)malloc(8 * n + 64);
    if (!work) return ORACLE_BAD_STRUCTURE;
    double error = SUF(gs_error)(n, ip, ix, dt, x, rhs, work);                         /* heat.rs:111 */
    for (uint64_t it = 0; it < max_iter; ++it) {
        for (uint64_t row = 0; row < n; ++row) {                                       /* outer_iterator().enumerate() */
            double sigma = 0.0, diag = 0.0;
            int have = 0;
            for (uint64_t p = (uint64_t)ip[row]; p < (uint64_t)ip[row + 1]; ++p) {
                const uint64_t col = (uint64_t)ix[p];
                if (col != row) {
                    const double prod = dt[p] * x[col];
                    sigma = sigma + prod;                                              /* sigma += val * x[[col_ind]] */
                } else {
                    diag = dt[p];
                    have = 1;
                }
            }
            if (!have) { free(work); return ORACLE_BAD_STRUCTURE; }                    /* diag.unwrap() */
            x[row] = (rhs[row] - sigma) / diag;
        }
        error = SUF(gs_error)(n, ip, ix, dt, x, rhs, work);
        if (error < eps) {
            info->iterations = it; info->error = error; info->converged = 1;
            free(work);
            return ORACLE_OK;
        }
    }
    info->iterations = max_iter; info->error = error; info->converged = 0;
    free(work);
    return ORACLE_OK;
}
