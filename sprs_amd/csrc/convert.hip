// Storage conversion and outer slicing on the device.
//
//   to_other_storage  — twin of raw::convert_mat_storage / CsMatBase::to_other_storage
//                       (sprs/src/sparse/csmat.rs:1405-1426, 1782-1829): CSR(outer x inner) ->
//                       the arrays of the other storage order (= CSR arrays of the transpose).
//                       The reference is a serial counting sort (histogram, cumsum, ordered fill);
//                       here: histogram (integer atomics) -> exclusive scan (scan.hip) -> unordered
//                       bucket fill -> every output row sorted by its (distinct) outer indices:
//                       rows up to 1024 entries by one wave (bitonic sort in LDS), longer rows by a
//                       workgroup with an LDS bitmap over windows of the outer range whose popcount
//                       prefix IS the rank (no comparison sort).  Output is therefore identical to
//                       the reference's: each output row strictly increasing.
//   slice_outer       — twin of slice_outer (sprs/src/sparse/slicing.rs:65-89) + to_proper
//                       (indptr.rs:206-214): a materialised, rebased copy of outer slices [start,end).
// Integer/HBM-bound; no MFMA.
#include "common.hpp"
#include "scan.hpp"

namespace sprs_hip {

namespace {

constexpr int WAVE = 64;
constexpr uint32_t KEY_PAD = 0xFFFFFFFFu;
constexpr int SMALL_ROW = 1024;          // longest row sorted by one wave
constexpr int CV_BLOCK = 256;
constexpr int CV_WAVES = CV_BLOCK / WAVE;
constexpr int WIN_LOG2 = 19;
constexpr uint64_t WIN = 1ull << WIN_LOG2;
constexpr int WORDS = (int)(WIN / 64);   // 8192 words, 64 KiB
constexpr int LG_BLOCK = 512;
constexpr int WORDS_PER_THREAD = WORDS / LG_BLOCK;

template <typename IDX>
__global__ void histogram_kernel(const IDX *__restrict__ indices, uint64_t nnz, unsigned long long *__restrict__ cnt) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride)
        atomicAdd(&cnt[(uint64_t)indices[p]], 1ull);
}

// bucket fill, order inside a bucket arbitrary (fixed afterwards); keys = outer index as u32
template <typename IDX, typename PTR>
__global__ __launch_bounds__(CV_BLOCK) void bucket_fill_kernel(const PTR *__restrict__ indptr,
                                                               const IDX *__restrict__ indices,
                                                               const double *__restrict__ data, uint64_t outer,
                                                               const uint64_t *__restrict__ o_ptr,
                                                               unsigned long long *__restrict__ cursor,
                                                               uint32_t *__restrict__ t_keys, double *__restrict__ t_vals) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t w0 = ((uint64_t)blockIdx.x * CV_BLOCK + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * CV_WAVES;
    for (uint64_t r = w0; r < outer; r += nw) {
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        for (uint64_t p = s + lane; p < e; p += WAVE) {
            const uint64_t j = (uint64_t)indices[p];
            const uint64_t pos = o_ptr[j] + atomicAdd(&cursor[j], 1ull);
            t_keys[pos] = (uint32_t)r;
            t_vals[pos] = data[p];
        }
    }
}

// rows with <= SMALL_ROW entries: one wave, bitonic sort by key in LDS
template <typename IDX>
__global__ __launch_bounds__(CV_BLOCK) void sort_small_rows_kernel(const uint64_t *__restrict__ o_ptr, uint64_t inner,
                                                                   const uint32_t *__restrict__ t_keys,
                                                                   const double *__restrict__ t_vals,
                                                                   IDX *__restrict__ o_indices,
                                                                   double *__restrict__ o_data) {
    __shared__ uint32_t keys_s[CV_WAVES][SMALL_ROW];
    __shared__ double vals_s[CV_WAVES][SMALL_ROW];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    uint32_t *keys = keys_s[wave];
    double *vals = vals_s[wave];
    const uint64_t w0 = (uint64_t)blockIdx.x * CV_WAVES + wave;
    const uint64_t nw = (uint64_t)gridDim.x * CV_WAVES;
    for (uint64_t j = w0; j < inner; j += nw) {
        const uint64_t s = o_ptr[j], e = o_ptr[j + 1];
        const uint32_t n = (uint32_t)(e - s);
        if (n == 0 || e - s > (uint64_t)SMALL_ROW) continue;
        if (n == 1) {
            if (lane == 0) {
                o_indices[s] = (IDX)t_keys[s];
                o_data[s] = t_vals[s];
            }
            continue;
        }
        uint32_t size = 2;
        while (size < n) size <<= 1;
        for (uint32_t i = lane; i < size; i += WAVE) {
            keys[i] = i < n ? t_keys[s + i] : KEY_PAD;
            vals[i] = i < n ? t_vals[s + i] : 0.0;
        }
        wave_sync_lds();
        for (uint32_t k2 = 2; k2 <= size; k2 <<= 1) {
            for (uint32_t jj = k2 >> 1; jj > 0; jj >>= 1) {
                for (uint32_t i = lane; i < size; i += WAVE) {
                    const uint32_t l = i ^ jj;
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
        for (uint32_t i = lane; i < n; i += WAVE) {
            o_indices[s + i] = (IDX)keys[i];
            o_data[s + i] = vals[i];
        }
        wave_sync_lds();
    }
}

// rows with > SMALL_ROW entries: one workgroup per row; the rank of a key inside the row is a
// popcount prefix over an LDS bitmap (keys are distinct), window by window over the outer range
template <typename IDX>
__global__ __launch_bounds__(LG_BLOCK) void sort_large_rows_kernel(const uint64_t *__restrict__ large_rows,
                                                                   const uint64_t *__restrict__ o_ptr, uint64_t outer,
                                                                   const uint32_t *__restrict__ t_keys,
                                                                   const double *__restrict__ t_vals,
                                                                   IDX *__restrict__ o_indices,
                                                                   double *__restrict__ o_data) {
    __shared__ unsigned long long bm[WORDS];
    __shared__ uint32_t wpre[WORDS];          // outputs before each word (within the window)
    __shared__ uint64_t wt[16];
    const uint32_t tid = threadIdx.x;
    const uint64_t j = large_rows[blockIdx.x];
    const uint64_t s = o_ptr[j], e = o_ptr[j + 1];
    uint64_t done = 0;                         // outputs placed by earlier windows
    for (uint64_t wlo = 0; wlo < outer; wlo += WIN) {
        const uint64_t whi = wlo + WIN;
        for (int i = tid; i < WORDS; i += LG_BLOCK) bm[i] = 0;
        __syncthreads();
        for (uint64_t p = s + tid; p < e; p += LG_BLOCK) {
            const uint64_t k = t_keys[p];
            if (k >= wlo && k < whi) atomicOr(&bm[(k - wlo) >> 6], 1ull << ((k - wlo) & 63));
        }
        __syncthreads();
        uint32_t mine = 0;
        uint32_t local[WORDS_PER_THREAD];
#pragma unroll
        for (int i = 0; i < WORDS_PER_THREAD; ++i) {
            local[i] = mine;
            mine += (uint32_t)__popcll(bm[tid * WORDS_PER_THREAD + i]);
        }
        uint64_t tot;
        const uint32_t tpre = (uint32_t)block_excl_scan_u64(mine, wt, &tot);
#pragma unroll
        for (int i = 0; i < WORDS_PER_THREAD; ++i) wpre[tid * WORDS_PER_THREAD + i] = tpre + local[i];
        __syncthreads();
        if (tot) {
            for (uint64_t p = s + tid; p < e; p += LG_BLOCK) {
                const uint64_t k = t_keys[p];
                if (k >= wlo && k < whi) {
                    const uint64_t c = k - wlo;
                    const uint32_t word = (uint32_t)(c >> 6);
                    const uint64_t rank = done + wpre[word] + (uint64_t)__popcll(bm[word] & ((1ull << (c & 63)) - 1ull));
                    o_indices[s + rank] = (IDX)k;
                    o_data[s + rank] = t_vals[p];
                }
            }
        }
        done += tot;
        __syncthreads();
    }
}

__global__ void find_large_rows_kernel(const uint64_t *__restrict__ o_ptr, uint64_t inner,
                                       uint64_t *__restrict__ large_rows, unsigned long long *__restrict__ n_large) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= inner) return;
    if (o_ptr[j + 1] - o_ptr[j] > (uint64_t)SMALL_ROW) large_rows[atomicAdd(n_large, 1ull)] = j;
}

template <typename PTR>
__global__ void narrow_ptr_kernel(const uint64_t *__restrict__ in, uint64_t n, uint64_t base, PTR *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (PTR)(in[i] - base);
}

template <typename PTR>
__global__ void rebase_ptr_kernel(const PTR *__restrict__ in, uint64_t n, PTR *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (PTR)(in[i] - in[0]);
}

struct Tmp {
    void *p = nullptr;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T>
    T *as() { return (T *)p; }
};

template <typename IDX, typename PTR>
int32_t convert_impl(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    hipStream_t stream = nullptr;
    const uint64_t outer = m->outer(), inner = m->inner(), nnz = m->nnz;
    // the reference tests mat.rows() against I whatever the storage (csmat.rs:1794-1797)
    if (sizeof(IDX) == 4 && m->rows > 0xFFFFFFFFull)
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW,
                  "Index type is not large enough to hold the number of rows requested (required %llu)",
                  (unsigned long long)m->rows);
    if (outer > 0xFFFFFFFEull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "to_other_storage: outer dimension above 2^32-2 is not supported");
    sprs_hip_csmat *o = nullptr;
    SPRS_TRY(alloc_csmat(&o, m->storage == SPRS_HIP_CSR ? SPRS_HIP_CSC : SPRS_HIP_CSR, m->rows, m->cols, nnz,
                         (int32_t)sizeof(PTR), (int32_t)sizeof(IDX)));
    struct Guard {
        sprs_hip_csmat *h;
        ~Guard() {
            if (h) sprs_hip_csmat_free(h);
        }
    } guard{o};
    Tmp cnt, optr, cursor, tkeys, tvals, large, nlarge;
    SPRS_TRY_HIP(cnt.alloc(inner * 8));
    SPRS_TRY_HIP(optr.alloc((inner + 1) * 8));
    SPRS_TRY_HIP(cursor.alloc(inner * 8));
    SPRS_TRY_HIP(tkeys.alloc(nnz * 4));
    SPRS_TRY_HIP(tvals.alloc(nnz * 8));
    SPRS_TRY_HIP(large.alloc(inner * 8));
    SPRS_TRY_HIP(nlarge.alloc(8));
    SPRS_TRY_HIP(hipMemsetAsync(cnt.p, 0, inner * 8 ? inner * 8 : 16, stream));
    SPRS_TRY_HIP(hipMemsetAsync(cursor.p, 0, inner * 8 ? inner * 8 : 16, stream));
    SPRS_TRY_HIP(hipMemsetAsync(nlarge.p, 0, 8, stream));
    if (nnz) {
        uint64_t blocks = (nnz + 255) / 256;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(histogram_kernel<IDX>, dim3((unsigned)blocks), dim3(256), 0, stream, (const IDX *)m->indices,
                           nnz, cnt.as<unsigned long long>());
        SPRS_TRY_HIP(hipGetLastError());
    }
    SPRS_TRY(exclusive_scan_u64(cnt.as<uint64_t>(), optr.as<uint64_t>(), inner, stream));
    hipLaunchKernelGGL(narrow_ptr_kernel<PTR>, dim3((unsigned)((inner + 256) / 256)), dim3(256), 0, stream,
                       optr.as<uint64_t>(), inner + 1, 0ull, (PTR *)o->indptr);
    if (nnz) {
        uint64_t wblocks = (outer + CV_WAVES - 1) / CV_WAVES;
        if (wblocks > 256 * 64) wblocks = 256 * 64;
        hipLaunchKernelGGL((bucket_fill_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(CV_BLOCK), 0, stream,
                           (const PTR *)m->indptr, (const IDX *)m->indices, m->data, outer, optr.as<uint64_t>(),
                           cursor.as<unsigned long long>(), tkeys.as<uint32_t>(), tvals.as<double>());
        uint64_t sblocks = (inner + CV_WAVES - 1) / CV_WAVES;
        if (sblocks > 256 * 32) sblocks = 256 * 32;
        hipLaunchKernelGGL(sort_small_rows_kernel<IDX>, dim3((unsigned)sblocks), dim3(CV_BLOCK), 0, stream,
                           optr.as<uint64_t>(), inner, tkeys.as<uint32_t>(), tvals.as<double>(), (IDX *)o->indices,
                           o->data);
        hipLaunchKernelGGL(find_large_rows_kernel, dim3((unsigned)((inner + 255) / 256)), dim3(256), 0, stream,
                           optr.as<uint64_t>(), inner, large.as<uint64_t>(), nlarge.as<unsigned long long>());
        SPRS_TRY_HIP(hipGetLastError());
        uint64_t n_large = 0;
        SPRS_TRY_HIP(hipMemcpy(&n_large, nlarge.p, 8, hipMemcpyDeviceToHost));
        if (n_large) {
            hipLaunchKernelGGL(sort_large_rows_kernel<IDX>, dim3((unsigned)n_large), dim3(LG_BLOCK), 0, stream,
                               large.as<uint64_t>(), optr.as<uint64_t>(), outer, tkeys.as<uint32_t>(),
                               tvals.as<double>(), (IDX *)o->indices, o->data);
            SPRS_TRY_HIP(hipGetLastError());
        }
    }
    SPRS_TRY_HIP(hipStreamSynchronize(stream));
    guard.h = nullptr;
    *out = o;
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR>
int32_t slice_impl(const sprs_hip_csmat *m, uint64_t start, uint64_t end, sprs_hip_csmat **out) {
    hipStream_t stream = nullptr;
    PTR lo = 0, hi = 0;
    SPRS_TRY_HIP(hipMemcpy(&lo, (const PTR *)m->indptr + start, sizeof(PTR), hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&hi, (const PTR *)m->indptr + end, sizeof(PTR), hipMemcpyDeviceToHost));
    const uint64_t nnz = (uint64_t)hi - (uint64_t)lo, n = end - start;
    sprs_hip_csmat *o = nullptr;
    const bool csr = m->storage == SPRS_HIP_CSR;
    SPRS_TRY(alloc_csmat(&o, m->storage, csr ? n : m->rows, csr ? m->cols : n, nnz, (int32_t)sizeof(PTR),
                         (int32_t)sizeof(IDX)));
    hipLaunchKernelGGL(rebase_ptr_kernel<PTR>, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, stream,
                       (const PTR *)m->indptr + start, n + 1, (PTR *)o->indptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && nnz)
        e = hipMemcpyAsync(o->indices, (const IDX *)m->indices + lo, nnz * sizeof(IDX), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess && nnz)
        e = hipMemcpyAsync(o->data, m->data + lo, nnz * sizeof(double), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        sprs_hip_csmat_free(o);
        return fail_hip(e, "slice_outer");
    }
    *out = o;
    return SPRS_HIP_OK;
}

}  // namespace

int32_t to_other_storage(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    if (m->idx_bytes == 8 && m->iptr_bytes == 8) return convert_impl<uint64_t, uint64_t>(m, out);
    if (m->idx_bytes == 4 && m->iptr_bytes == 8) return convert_impl<uint32_t, uint64_t>(m, out);
    if (m->idx_bytes == 8 && m->iptr_bytes == 4) return convert_impl<uint64_t, uint32_t>(m, out);
    return convert_impl<uint32_t, uint32_t>(m, out);
}

int32_t slice_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end, sprs_hip_csmat **out) {
    if (m->idx_bytes == 8 && m->iptr_bytes == 8) return slice_impl<uint64_t, uint64_t>(m, start, end, out);
    if (m->idx_bytes == 4 && m->iptr_bytes == 8) return slice_impl<uint32_t, uint64_t>(m, start, end, out);
    if (m->idx_bytes == 8 && m->iptr_bytes == 4) return slice_impl<uint64_t, uint32_t>(m, start, end, out);
    return slice_impl<uint32_t, uint32_t>(m, start, end, out);
}

}  // namespace sprs_hip
