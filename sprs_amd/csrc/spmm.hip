// CSR x dense (row-major) SpMM for gfx950 — device twin of
//   prod::csr_mulacc_dense_rowmaj   sprs/src/sparse/prod.rs:189-214
// (what `&CsMat * &Array2` dispatches to when the rhs has >= 8 columns, csmat.rs:2002-2016):
//     out[i, :] += a_ik * rhs[k, :]     for every stored a_ik, ascending k.
// First "next" row of SURVEY §8(f): the same CSR stream as SpMV, but every gathered x entry
// becomes a whole ROW of the rhs — k contiguous doubles — so one L2 line access now carries
// k/16 .. 1 full line of useful data instead of 8 bytes of it.
//
// Decomposition (plan cached in the handle):
//  * DEFAULT: every row is cut into CHUNKS of 512 entries, one wave per chunk (lane j of group g takes entries g, g + G,
//    ... of the chunk, groups of KP lanes — KP = k rounded up to a power of two — combined with xor-shuffles), partials of
//    multi-chunk rows added in chunk order by a second kernel: deterministic, equal to the reference up to the summation
//    order (tests: 1e-12 relative).  All 64 lanes stream entries and every lane has its gathers in flight: 12.1 ms at
//    k = 16 on R-MAT 10M (3.4 TB/s of rhs-row gathers — the bound: 41 GB of 128-byte rows from a 1.28 GB operand).
//  * option spmm_long_row = L > 0: rows of at most L entries instead belong to a GROUP of KP lanes that walks its rows
//    as a little state machine (32 entries and their rhs rows in flight per turn) and adds the products IN ENTRY ORDER
//    into one accumulator per column — the reference's own order, bit for bit, the accumulate form included.  Measured
//    3.4x slower (41 ms at k = 16, profiles/r03f: the serial chains and 64-bit shuffles leave the gathers idle), hence
//    opt-in for callers that need the reference's bits.
// Unfused multiply-add (-ffp-contract=off) like MulAcc (mul_acc.rs:28-30).
#include "common.hpp"
#include "scan.hpp"

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr int MM_BLOCK = 256;
constexpr int MM_WAVES = MM_BLOCK / WAVE;
constexpr uint64_t CHUNK = 512;
constexpr uint64_t LONG_ROW = 0;           // default of option spmm_long_row: rows above this many entries go to the chunk kernels (0: all)

template <typename PTR>
__global__ void count_chunks_kernel(const PTR *__restrict__ indptr, uint64_t rows, uint64_t long_row, uint64_t *__restrict__ nchunks,
                                    uint64_t *__restrict__ is_multi) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t len = (uint64_t)indptr[r + 1] - (uint64_t)indptr[r];
    const uint64_t n = len > long_row ? (len + CHUNK - 1) / CHUNK : 0;
    nchunks[r] = n;
    is_multi[r] = n > 1 ? 1 : 0;
}

__global__ void fill_chunks_kernel(const uint64_t *__restrict__ first_chunk, const uint64_t *__restrict__ multi_pos,
                                   uint64_t rows, uint64_t *__restrict__ chunk_row, uint64_t *__restrict__ multi_rows) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t f = first_chunk[r], n = first_chunk[r + 1] - f;
    for (uint64_t c = 0; c < n; ++c) chunk_row[f + c] = r;
    if (n > 1) multi_rows[multi_pos[r]] = r;
}

// rows of at most LONG_ROW entries: one group of KP lanes per row at a time (see the header)
template <typename IDX, typename PTR, int KP, bool ACC>
__global__ __launch_bounds__(MM_BLOCK) void spmm_rows_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                             const double *__restrict__ data, uint64_t rows,
                                                             const double *__restrict__ rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k,
                                                             double *__restrict__ out, uint64_t ld_out, uint64_t cs_out, uint64_t long_row) {
    constexpr int EB = KP <= 16 ? 32 / KP : 1;           // entries a lane loads per turn
    constexpr int NB = EB * KP;                          // entries of a turn (32, or KP)
    constexpr int UN = NB < 32 ? NB : 32;                // rhs rows in flight per group
    const uint32_t j = threadIdx.x % KP;
    const bool col_ok = j < k;
    const uint64_t ng = (uint64_t)gridDim.x * (MM_BLOCK / KP);
    uint64_t r = ((uint64_t)blockIdx.x * MM_BLOCK + threadIdx.x) / KP;
    bool active = r < rows, mine = false;
    uint64_t cur = 0, end = 0, ncur = 0, nend = 0;
    double acc = 0.0;
    if (active) {
        cur = (uint64_t)indptr[r];
        end = (uint64_t)indptr[r + 1];
        mine = end - cur <= long_row;
        if (!mine) cur = end;
        if constexpr (ACC)
            if (mine && col_ok) acc = out[r * ld_out + j * cs_out];
        if (r + ng < rows) {
            ncur = (uint64_t)indptr[r + ng];
            nend = (uint64_t)indptr[r + ng + 1];
        }
    }
    while (__ballot(active) != 0ull) {
        const uint32_t nb = active ? (uint32_t)(end - cur < (uint64_t)NB ? end - cur : (uint64_t)NB) : 0u;
        uint64_t col[EB];
        double val[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const uint32_t t = (uint32_t)e * KP + j;
            const bool ok = t < nb;
            col[e] = ok ? (uint64_t)indices[cur + t] : 0ull;
            val[e] = ok ? data[cur + t] : 0.0;
        }
#pragma unroll
        for (int t0 = 0; t0 < NB; t0 += UN) {
            if (__ballot((uint32_t)t0 < nb) == 0ull) break;            // wave-uniform
            double x[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int t = t0 + u;
                // (an entry past the end reads rhs row 0 and is not added: no branch in the load sequence)
                const uint64_t c = __shfl(col[t / KP], t % KP, KP);
                x[u] = col_ok ? rhs[c * ld_rhs + j * cs_rhs] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int t = t0 + u;
                const double v = __shfl(val[t / KP], t % KP, KP);
                const double sum = acc + v * x[u];
                acc = (uint32_t)t < nb ? sum : acc;
            }
        }
        cur += nb;
        if (active && cur == end) {
            if (mine && col_ok) out[r * ld_out + j * cs_out] = acc;
            r += ng;
            active = r < rows;
            cur = ncur;
            end = nend;
            acc = 0.0;
            if (active) {
                mine = end - cur <= long_row;
                if (!mine) cur = end;
                if constexpr (ACC)
                    if (mine && col_ok) acc = out[r * ld_out + j * cs_out];
                if (r + ng < rows) {
                    ncur = (uint64_t)indptr[r + ng];
                    nend = (uint64_t)indptr[r + ng + 1];
                }
            }
        }
    }
}

template <typename IDX, typename PTR, int KP, bool ACC>
__global__ __launch_bounds__(MM_BLOCK) void spmm_chunk_kernel(const PTR *__restrict__ indptr,
                                                              const IDX *__restrict__ indices,
                                                              const double *__restrict__ data,
                                                              const uint64_t *__restrict__ chunk_row,
                                                              const uint64_t *__restrict__ first_chunk,
                                                              uint64_t nchunks, const double *__restrict__ rhs,
                                                              uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k, double *__restrict__ out,
                                                              uint64_t ld_out, uint64_t cs_out, double *__restrict__ partial) {
    constexpr int G = WAVE / KP;                       // entries processed concurrently by one wave
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t j = lane % KP, g = lane / KP;
    const uint64_t w0 = ((uint64_t)blockIdx.x * MM_BLOCK + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * MM_WAVES;
    for (uint64_t c = w0; c < nchunks; c += nw) {
        const uint64_t r = chunk_row[c];
        const uint64_t f = first_chunk[r], nc = first_chunk[r + 1] - f;
        const uint64_t rs = (uint64_t)indptr[r], re = (uint64_t)indptr[r + 1];
        const uint64_t s = rs + (c - f) * CHUNK;
        const uint64_t e = (s + CHUNK < re) ? s + CHUNK : re;
        double acc = 0.0;
        if (j < k) {
            for (uint64_t p = s + g; p < e; p += 4 * G) {              // four rhs rows in flight; added in the order g, g + G, ...
                uint64_t col[4];
                double a[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = p + (uint64_t)u * G < e;
                    col[u] = ok ? (uint64_t)indices[p + (uint64_t)u * G] : 0ull;
                    a[u] = ok ? data[p + (uint64_t)u * G] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = rhs[col[u] * ld_rhs + j * cs_rhs];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double sum = acc + a[u] * x[u];
                    acc = p + (uint64_t)u * G < e ? sum : acc;
                }
            }
        }
#pragma unroll
        for (int o = KP; o < WAVE; o <<= 1) acc += __shfl_xor(acc, o, WAVE);
        if (g == 0 && j < k) {
            if (nc == 1) {
                double *dst = out + r * ld_out + j * cs_out;
                if constexpr (ACC) *dst = *dst + acc;
                else *dst = acc;
            } else {
                partial[c * (uint64_t)k + j] = acc;
            }
        }
    }
}

// rows longer than one chunk: add the chunk partials in chunk order
template <bool ACC>
__global__ void spmm_combine_kernel(const uint64_t *__restrict__ multi_rows, uint64_t n_multi,
                                    const uint64_t *__restrict__ first_chunk, const double *__restrict__ partial,
                                    uint32_t k, double *__restrict__ out, uint64_t ld_out, uint64_t cs_out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_multi * k) return;
    const uint64_t r = multi_rows[t / k];
    const uint32_t j = (uint32_t)(t % k);
    const uint64_t f = first_chunk[r], n = first_chunk[r + 1] - f;
    double s = 0.0;
    for (uint64_t c = 0; c < n; ++c) s += partial[(f + c) * (uint64_t)k + j];
    double *dst = out + r * ld_out + j * cs_out;
    if constexpr (ACC) *dst = *dst + s;
    else *dst = s;
}

struct Tmp {
    void *p = nullptr;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    uint64_t *u64() { return (uint64_t *)p; }
};

template <typename PTR>
int32_t build_spmm_plan(sprs_hip_csmat *a, hipStream_t stream) {
    SpmmPlan &pl = a->mm;
    pl.release();
    pl.long_row = options().spmm_long_row >= 0 ? (uint64_t)options().spmm_long_row : LONG_ROW;
    const uint64_t rows = a->rows;
    Tmp nch, mflag, mpos;
    SPRS_TRY_HIP(nch.alloc(rows * 8));
    SPRS_TRY_HIP(mflag.alloc(rows * 8));
    SPRS_TRY_HIP(mpos.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&pl.first_chunk, (rows + 1) * 8));
    hipLaunchKernelGGL(count_chunks_kernel<PTR>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream,
                       (const PTR *)a->indptr, rows, pl.long_row, nch.u64(), mflag.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(nch.u64(), pl.first_chunk, rows, stream));
    SPRS_TRY(exclusive_scan_u64(mflag.u64(), mpos.u64(), rows, stream));
    SPRS_TRY_HIP(hipMemcpy(&pl.nchunks, pl.first_chunk + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&pl.n_multi, mpos.u64() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMalloc((void **)&pl.chunk_row, (pl.nchunks ? pl.nchunks : 1) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&pl.multi_rows, (pl.n_multi ? pl.n_multi : 1) * 8));
    hipLaunchKernelGGL(fill_chunks_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, pl.first_chunk,
                       mpos.u64(), rows, pl.chunk_row, pl.multi_rows);
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY_HIP(hipStreamSynchronize(stream));
    pl.built = true;
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR, int KP>
int32_t launch_block(sprs_hip_csmat *a, const double *rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k, double *out, uint64_t ld_out,
                     uint64_t cs_out, bool acc, double *partial, hipStream_t stream) {
    const SpmmPlan &pl = a->mm;
    {
        // short rows: groups of KP lanes, group-stride; enough workgroups to fill the chip eight deep
        constexpr uint64_t groups_per_block = MM_BLOCK / KP;
        uint64_t blocks = (a->rows + groups_per_block - 1) / groups_per_block;
        if (blocks > 256 * 8) blocks = 256 * 8;
        const dim3 grid((unsigned)blocks), block(MM_BLOCK);
        if (acc)
            hipLaunchKernelGGL((spmm_rows_kernel<IDX, PTR, KP, true>), grid, block, 0, stream, (const PTR *)a->indptr,
                               (const IDX *)a->indices, a->data, a->rows, rhs, ld_rhs, cs_rhs, k, out, ld_out, cs_out, pl.long_row);
        else
            hipLaunchKernelGGL((spmm_rows_kernel<IDX, PTR, KP, false>), grid, block, 0, stream, (const PTR *)a->indptr,
                               (const IDX *)a->indices, a->data, a->rows, rhs, ld_rhs, cs_rhs, k, out, ld_out, cs_out, pl.long_row);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (!pl.nchunks) return SPRS_HIP_OK;
    uint64_t blocks = (pl.nchunks + MM_WAVES - 1) / MM_WAVES;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const dim3 grid((unsigned)blocks), block(MM_BLOCK);
    if (acc)
        hipLaunchKernelGGL((spmm_chunk_kernel<IDX, PTR, KP, true>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, pl.chunk_row, pl.first_chunk, pl.nchunks, rhs, ld_rhs, cs_rhs, k,
                           out, ld_out, cs_out, partial);
    else
        hipLaunchKernelGGL((spmm_chunk_kernel<IDX, PTR, KP, false>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, pl.chunk_row, pl.first_chunk, pl.nchunks, rhs, ld_rhs, cs_rhs, k,
                           out, ld_out, cs_out, partial);
    SPRS_TRY_HIP(hipGetLastError());
    if (pl.n_multi) {
        const uint64_t th = pl.n_multi * k;
        const dim3 g2((unsigned)((th + 255) / 256)), b2(256);
        if (acc)
            hipLaunchKernelGGL(spmm_combine_kernel<true>, g2, b2, 0, stream, pl.multi_rows, pl.n_multi, pl.first_chunk,
                               partial, k, out, ld_out, cs_out);
        else
            hipLaunchKernelGGL(spmm_combine_kernel<false>, g2, b2, 0, stream, pl.multi_rows, pl.n_multi, pl.first_chunk,
                               partial, k, out, ld_out, cs_out);
        SPRS_TRY_HIP(hipGetLastError());
    }
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR>
int32_t spmm_impl(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t ld_rhs, uint64_t cs_rhs, double *out, uint64_t ld_out,
                  uint64_t cs_out, bool acc, hipStream_t stream) {
    double *partial = nullptr;
    std::lock_guard<std::recursive_mutex> lock(a->mu);   // held until the kernels that read the plan are launched
    {
        const uint64_t want_long = options().spmm_long_row >= 0 ? (uint64_t)options().spmm_long_row : LONG_ROW;
        if (!a->mm.built || a->mm.long_row != want_long) SPRS_TRY(build_spmm_plan<PTR>(a, stream));
        SpmmPlan &pl = a->mm;
        const uint64_t kb = k < 64 ? k : 64;
        const uint64_t need = (pl.n_multi ? pl.nchunks : 0) * kb * sizeof(double);
        auto &slot = pl.partial[(void *)stream];
        if (slot.second < need) {
            if (slot.first) (void)hipFree(slot.first);
            slot.first = nullptr;
            slot.second = 0;
            SPRS_TRY_HIP(hipMalloc((void **)&slot.first, need));
            slot.second = need;
        }
        partial = slot.first;
    }
    if (a->nnz == 0) {
        // nothing stored: the operator form is all zeros (csmat.rs:2004), the accumulate form leaves `out` alone
        if (!acc) {
            if (cs_out != 1) SPRS_TRY_HIP(hipMemset2DAsync(out, cs_out * sizeof(double), 0, a->rows * sizeof(double), k, stream));   // column-major: k columns of `rows` doubles
            else if (ld_out == k) SPRS_TRY_HIP(hipMemsetAsync(out, 0, a->rows * k * sizeof(double), stream));
            else SPRS_TRY_HIP(hipMemset2DAsync(out, ld_out * sizeof(double), 0, k * sizeof(double), a->rows, stream));
        }
        return SPRS_HIP_OK;
    }
    // every row is written by exactly one kernel (an empty row as zeros): no memset of `out`
    for (uint64_t j0 = 0; j0 < k; j0 += 64) {          // column blocks of 64
        const uint32_t kb = (uint32_t)(k - j0 < 64 ? k - j0 : 64);
        const double *r = rhs + j0 * cs_rhs;
        double *o = out + j0 * cs_out;
        int32_t st;
        if (kb <= 8) st = launch_block<IDX, PTR, 8>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, stream);
        else if (kb <= 16) st = launch_block<IDX, PTR, 16>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, stream);
        else if (kb <= 32) st = launch_block<IDX, PTR, 32>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, stream);
        else st = launch_block<IDX, PTR, 64>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, stream);
        SPRS_TRY(st);
    }
    return SPRS_HIP_OK;
}

}  // namespace

void SpmmPlan::release() {
    auto drop = [](void *p) {
        if (p) (void)hipFree(p);
    };
    drop(first_chunk);
    drop(chunk_row);
    drop(multi_rows);
    first_chunk = chunk_row = multi_rows = nullptr;
    for (auto &kv : partial) drop(kv.second.first);
    partial.clear();
    nchunks = n_multi = 0;
    built = false;
}

// General strides: element (r, j) of the rhs at rhs[r * rs_rhs + j * cs_rhs], of out at out[r * rs_out + j * cs_out] — row-major
// operands have cs = 1, column-major ones rs = 1.  One kernel family serves the four layout pairs `&CsMat * &Array2` can meet
// (csmat.rs:1989-2048: the result is row-major for >= 8 columns, column-major below, whatever the rhs is).
int32_t spmm_strided_f64(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t rs_rhs, uint64_t cs_rhs, double *out,
                         uint64_t rs_out, uint64_t cs_out, bool accumulate, hipStream_t stream) {
    if (a->rows == 0 || k == 0) return SPRS_HIP_OK;
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return spmm_impl<uint64_t, uint64_t>(a, rhs, k, rs_rhs, cs_rhs, out, rs_out, cs_out, accumulate, stream);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return spmm_impl<uint32_t, uint64_t>(a, rhs, k, rs_rhs, cs_rhs, out, rs_out, cs_out, accumulate, stream);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return spmm_impl<uint64_t, uint32_t>(a, rhs, k, rs_rhs, cs_rhs, out, rs_out, cs_out, accumulate, stream);
    return spmm_impl<uint32_t, uint32_t>(a, rhs, k, rs_rhs, cs_rhs, out, rs_out, cs_out, accumulate, stream);
}

int32_t spmm_rowmaj_f64(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t ld_rhs, double *out,
                        uint64_t ld_out, bool accumulate, hipStream_t stream) {
    return spmm_strided_f64(a, rhs, k, ld_rhs, 1, out, ld_out, 1, accumulate, stream);
}

}  // namespace sprs_hip
