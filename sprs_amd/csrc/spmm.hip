// CSR x dense (row-major) SpMM for gfx950 — device twin of
//   prod::csr_mulacc_dense_rowmaj   sprs/src/sparse/prod.rs:189-214
// (what `&CsMat * &Array2` dispatches to when the rhs has >= 8 columns, csmat.rs:2002-2016):
//     out[i, :] += a_ik * rhs[k, :]     for every stored a_ik, ascending k.
// First "next" row of SURVEY §8(f): the same CSR stream as SpMV, but every gathered x entry
// becomes a whole ROW of the rhs — k contiguous doubles — so one L2 line access now carries
// k/16 .. 1 full line of useful data instead of 8 bytes of it.
//
// Decomposition: rows are cut into CHUNKS of at most 512 entries (plan, cached in the handle).
// One wave per chunk; the 64 lanes are 64/KP groups of KP lanes (KP = k rounded up to a power of
// two): lane j of group g accumulates column j over entries g, g + G, ... of the chunk, the rhs row
// of an entry is read by the KP lanes of a group as one contiguous segment.  Groups are combined
// with xor-shuffles.  A row that fits one chunk is written directly; the chunks of longer rows go
// to a scratch array and are added in chunk order by a second kernel (no float atomics:
// deterministic).  Unfused multiply-add (-ffp-contract=off) like MulAcc (mul_acc.rs:28-30).
#include "common.hpp"
#include "scan.hpp"

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr int MM_BLOCK = 256;
constexpr int MM_WAVES = MM_BLOCK / WAVE;
constexpr uint64_t CHUNK = 512;

template <typename PTR>
__global__ void count_chunks_kernel(const PTR *__restrict__ indptr, uint64_t rows, uint64_t *__restrict__ nchunks,
                                    uint64_t *__restrict__ is_multi) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t len = (uint64_t)indptr[r + 1] - (uint64_t)indptr[r];
    const uint64_t n = (len + CHUNK - 1) / CHUNK;
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

template <typename IDX, typename PTR, int KP, bool ACC>
__global__ __launch_bounds__(MM_BLOCK) void spmm_chunk_kernel(const PTR *__restrict__ indptr,
                                                              const IDX *__restrict__ indices,
                                                              const double *__restrict__ data,
                                                              const uint64_t *__restrict__ chunk_row,
                                                              const uint64_t *__restrict__ first_chunk,
                                                              uint64_t nchunks, const double *__restrict__ rhs,
                                                              uint64_t ld_rhs, uint32_t k, double *__restrict__ out,
                                                              uint64_t ld_out, double *__restrict__ partial) {
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
            for (uint64_t p = s + g; p < e; p += G) {
                const uint64_t col = (uint64_t)indices[p];
                const double prod = data[p] * rhs[col * ld_rhs + j];
                acc += prod;
            }
        }
#pragma unroll
        for (int o = KP; o < WAVE; o <<= 1) acc += __shfl_xor(acc, o, WAVE);
        if (g == 0 && j < k) {
            if (nc == 1) {
                double *dst = out + r * ld_out + j;
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
                                    uint32_t k, double *__restrict__ out, uint64_t ld_out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_multi * k) return;
    const uint64_t r = multi_rows[t / k];
    const uint32_t j = (uint32_t)(t % k);
    const uint64_t f = first_chunk[r], n = first_chunk[r + 1] - f;
    double s = 0.0;
    for (uint64_t c = 0; c < n; ++c) s += partial[(f + c) * (uint64_t)k + j];
    double *dst = out + r * ld_out + j;
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
    const uint64_t rows = a->rows;
    Tmp nch, mflag, mpos;
    SPRS_TRY_HIP(nch.alloc(rows * 8));
    SPRS_TRY_HIP(mflag.alloc(rows * 8));
    SPRS_TRY_HIP(mpos.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&pl.first_chunk, (rows + 1) * 8));
    hipLaunchKernelGGL(count_chunks_kernel<PTR>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream,
                       (const PTR *)a->indptr, rows, nch.u64(), mflag.u64());
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
int32_t launch_block(sprs_hip_csmat *a, const double *rhs, uint64_t ld_rhs, uint32_t k, double *out, uint64_t ld_out,
                     bool acc, double *partial, hipStream_t stream) {
    const SpmmPlan &pl = a->mm;
    uint64_t blocks = (pl.nchunks + MM_WAVES - 1) / MM_WAVES;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const dim3 grid((unsigned)blocks), block(MM_BLOCK);
    if (acc)
        hipLaunchKernelGGL((spmm_chunk_kernel<IDX, PTR, KP, true>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, pl.chunk_row, pl.first_chunk, pl.nchunks, rhs, ld_rhs, k,
                           out, ld_out, partial);
    else
        hipLaunchKernelGGL((spmm_chunk_kernel<IDX, PTR, KP, false>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, pl.chunk_row, pl.first_chunk, pl.nchunks, rhs, ld_rhs, k,
                           out, ld_out, partial);
    SPRS_TRY_HIP(hipGetLastError());
    if (pl.n_multi) {
        const uint64_t th = pl.n_multi * k;
        const dim3 g2((unsigned)((th + 255) / 256)), b2(256);
        if (acc)
            hipLaunchKernelGGL(spmm_combine_kernel<true>, g2, b2, 0, stream, pl.multi_rows, pl.n_multi, pl.first_chunk,
                               partial, k, out, ld_out);
        else
            hipLaunchKernelGGL(spmm_combine_kernel<false>, g2, b2, 0, stream, pl.multi_rows, pl.n_multi, pl.first_chunk,
                               partial, k, out, ld_out);
        SPRS_TRY_HIP(hipGetLastError());
    }
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR>
int32_t spmm_impl(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t ld_rhs, double *out, uint64_t ld_out,
                  bool acc, hipStream_t stream) {
    double *partial = nullptr;
    {
        std::lock_guard<std::mutex> lock(a->mu);
        if (!a->mm.built) SPRS_TRY(build_spmm_plan<PTR>(a, stream));
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
    const SpmmPlan &pl = a->mm;
    if (!acc) {
        // rows without a chunk (empty rows) must read as zero in the operator form (csmat.rs:2004)
        if (ld_out == k) SPRS_TRY_HIP(hipMemsetAsync(out, 0, a->rows * k * sizeof(double), stream));
        else SPRS_TRY_HIP(hipMemset2DAsync(out, ld_out * sizeof(double), 0, k * sizeof(double), a->rows, stream));
    }
    if (!pl.nchunks) return SPRS_HIP_OK;
    for (uint64_t j0 = 0; j0 < k; j0 += 64) {          // column blocks of 64
        const uint32_t kb = (uint32_t)(k - j0 < 64 ? k - j0 : 64);
        const double *r = rhs + j0;
        double *o = out + j0;
        int32_t st;
        if (kb <= 8) st = launch_block<IDX, PTR, 8>(a, r, ld_rhs, kb, o, ld_out, acc, partial, stream);
        else if (kb <= 16) st = launch_block<IDX, PTR, 16>(a, r, ld_rhs, kb, o, ld_out, acc, partial, stream);
        else if (kb <= 32) st = launch_block<IDX, PTR, 32>(a, r, ld_rhs, kb, o, ld_out, acc, partial, stream);
        else st = launch_block<IDX, PTR, 64>(a, r, ld_rhs, kb, o, ld_out, acc, partial, stream);
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

int32_t spmm_rowmaj_f64(sprs_hip_csmat *a, const double *rhs, uint64_t k, uint64_t ld_rhs, double *out,
                        uint64_t ld_out, bool accumulate, hipStream_t stream) {
    if (a->rows == 0 || k == 0) return SPRS_HIP_OK;
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return spmm_impl<uint64_t, uint64_t>(a, rhs, k, ld_rhs, out, ld_out, accumulate, stream);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return spmm_impl<uint32_t, uint64_t>(a, rhs, k, ld_rhs, out, ld_out, accumulate, stream);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return spmm_impl<uint64_t, uint32_t>(a, rhs, k, ld_rhs, out, ld_out, accumulate, stream);
    return spmm_impl<uint32_t, uint32_t>(a, rhs, k, ld_rhs, out, ld_out, accumulate, stream);
}

}  // namespace sprs_hip
