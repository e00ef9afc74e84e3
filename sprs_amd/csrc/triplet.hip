// Triplet (COO) -> CSR / CSC assembly on the device — twin of TriMatBase::to_csr / to_csc
// (sprs/src/sparse/triplet.rs:262-276) = TriMatIter::into_cs (sprs/src/sparse/triplet_iter.rs:127-224):
//   sort the triplets by (outer, inner)            triplet_iter.rs:150-158 (sort_unstable_by_key there)
//   fold equal neighbours, slot = slot + next      triplet_iter.rs:160-180, left to right
//   fill indptr, empty outer slices included       triplet_iter.rs:182-214
// Here the sort is a STABLE radix sort (sort.hip), so a group of duplicates is summed in triplet order — one of the
// orders the reference's unstable sort may produce (the CPU restatement used by the tests takes the same one).  Explicit zeros and
// sums that cancel stay stored.  HBM-bound integer work (32 B per triplet and radix pass), no MFMA.
#include "common.hpp"

#include <vector>

namespace sprs_hip {

int32_t radix_sort_pairs(uint64_t *keys, uint64_t *vals, uint64_t n, const std::vector<std::pair<int, int>> &fields, hipStream_t stream);   // sort.hip

namespace {

struct Buf {
    void *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    uint64_t *u64() { return (uint64_t *)p; }
};

template <typename I>
__global__ void tri_keys_kernel(const I *__restrict__ outer, const I *__restrict__ inner, const double *__restrict__ data, uint64_t n,
                                uint64_t n_outer, uint64_t n_inner, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals,
                                unsigned int *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t o = (uint64_t)outer[i], c = (uint64_t)inner[i];
    if (o >= n_outer || c >= n_inner) atomicOr(bad, 1u);       // add_triplet asserts the bounds (triplet.rs:171-172)
    keys[i] = (o << 32) | (c & 0xFFFFFFFFull);
    vals[i] = (uint64_t)__double_as_longlong(data[i]);
}

__global__ void tri_heads_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one thread per group of equal (outer, inner): the sum in sorted (= triplet) order, the index, and the indptr entries
// of the outer slices that begin at this group (the empty ones before it included)
template <typename I, typename P>
__global__ void tri_fold_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals, const uint64_t *__restrict__ head,
                                const uint64_t *__restrict__ gidx, uint64_t n, uint64_t n_outer, uint64_t ngroups,
                                P *__restrict__ indptr, I *__restrict__ indices, double *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const uint64_t g = gidx[i], key = keys[i];
    double acc = __longlong_as_double((long long)vals[i]);
    for (uint64_t q = i + 1; q < n && !head[q]; ++q) acc = acc + __longlong_as_double((long long)vals[q]);   // slot = slot + next
    indices[g] = (I)(key & 0xFFFFFFFFull);
    out[g] = acc;
    const uint64_t o = key >> 32;
    const uint64_t prev = i ? (keys[i - 1] >> 32) : ~0ull;      // outer slice of the previous group (none: -1)
    if (i == 0 || prev != o)
        for (uint64_t r = (i ? prev + 1 : 0); r <= o; ++r) indptr[r] = (P)g;
    if (g + 1 == ngroups)
        for (uint64_t r = o + 1; r <= n_outer; ++r) indptr[r] = (P)ngroups;
}

template <typename P>
__global__ void tri_empty_indptr_kernel(P *__restrict__ indptr, uint64_t n_outer) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_outer) indptr[r] = 0;
}

int bits_for(uint64_t n) {       // bits needed for values < n
    int b = 0;
    while (b < 32 && (1ull << b) < n) ++b;
    return b ? b : 1;
}

template <typename I>
int32_t assemble(uint64_t rows, uint64_t cols, uint64_t n, const I *row_inds, const I *col_inds, const double *data, int32_t storage,
                 int32_t out_idx_bytes, int32_t out_iptr_bytes, sprs_hip_csmat **out) {
    hipStream_t stream = nullptr;
    const bool csr = storage == SPRS_HIP_CSR;
    const uint64_t n_outer = csr ? rows : cols, n_inner = csr ? cols : rows;
    const I *outer = csr ? row_inds : col_inds, *inner = csr ? col_inds : row_inds;
    sprs_hip_csmat *c = nullptr;
    if (n == 0) {
        SPRS_TRY(alloc_csmat(&c, storage, rows, cols, 0, out_iptr_bytes, out_idx_bytes));
        if (out_iptr_bytes == 8) hipLaunchKernelGGL(tri_empty_indptr_kernel<uint64_t>, dim3((unsigned)((n_outer + 256) / 256)), dim3(256), 0, stream, (uint64_t *)c->indptr, n_outer);
        else hipLaunchKernelGGL(tri_empty_indptr_kernel<uint32_t>, dim3((unsigned)((n_outer + 256) / 256)), dim3(256), 0, stream, (uint32_t *)c->indptr, n_outer);
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            sprs_hip_csmat_free(c);
            return fail_hip(e, "triplets_to_cs");
        }
        *out = c;
        return SPRS_HIP_OK;
    }
    Buf keys, vals, head, gidx, bad;
    SPRS_TRY_HIP(keys.alloc(n * 8));
    SPRS_TRY_HIP(vals.alloc(n * 8));
    SPRS_TRY_HIP(head.alloc(n * 8));
    SPRS_TRY_HIP(gidx.alloc((n + 1) * 8));
    SPRS_TRY_HIP(bad.alloc(4));
    SPRS_TRY_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
    const dim3 g1((unsigned)((n + 255) / 256)), b1(256);
    hipLaunchKernelGGL(tri_keys_kernel<I>, g1, b1, 0, stream, outer, inner, data, n, n_outer, n_inner, keys.u64(), vals.u64(), (unsigned int *)bad.p);
    SPRS_TRY_HIP(hipGetLastError());
    unsigned int isbad = 0;
    SPRS_TRY_HIP(hipMemcpy(&isbad, bad.p, 4, hipMemcpyDeviceToHost));
    if (isbad) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "triplet index out of bounds");
    SPRS_TRY(radix_sort_pairs(keys.u64(), vals.u64(), n, {{0, bits_for(n_inner)}, {32, bits_for(n_outer)}}, stream));
    hipLaunchKernelGGL(tri_heads_kernel, g1, b1, 0, stream, (const uint64_t *)keys.u64(), n, head.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(head.u64(), gidx.u64(), n, stream));
    uint64_t ngroups = 0;
    SPRS_TRY_HIP(hipMemcpy(&ngroups, gidx.u64() + n, 8, hipMemcpyDeviceToHost));
    if (out_iptr_bytes == 4 && ngroups > 0xFFFFFFFFull)
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold the nnz of the matrix (%llu)", (unsigned long long)ngroups);
    SPRS_TRY(alloc_csmat(&c, storage, rows, cols, ngroups, out_iptr_bytes, out_idx_bytes));
#define SPRS_TRI_FOLD(IT, PT)                                                                                                   \
    hipLaunchKernelGGL((tri_fold_kernel<IT, PT>), g1, b1, 0, stream, (const uint64_t *)keys.u64(), (const uint64_t *)vals.u64(), \
                       (const uint64_t *)head.u64(), (const uint64_t *)gidx.u64(), n, n_outer, ngroups, (PT *)c->indptr,          \
                       (IT *)c->indices, c->data)
    if (out_idx_bytes == 8 && out_iptr_bytes == 8) SPRS_TRI_FOLD(uint64_t, uint64_t);
    else if (out_idx_bytes == 4 && out_iptr_bytes == 8) SPRS_TRI_FOLD(uint32_t, uint64_t);
    else if (out_idx_bytes == 8) SPRS_TRI_FOLD(uint64_t, uint32_t);
    else SPRS_TRI_FOLD(uint32_t, uint32_t);
#undef SPRS_TRI_FOLD
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        sprs_hip_csmat_free(c);
        return fail_hip(e, "triplets_to_cs");
    }
    *out = c;
    return SPRS_HIP_OK;
}

}  // namespace

int32_t triplets_to_cs(uint64_t rows, uint64_t cols, uint64_t n, const void *row_inds, const void *col_inds, int32_t in_idx_bytes,
                       const double *data, int32_t storage, int32_t out_idx_bytes, int32_t out_iptr_bytes, sprs_hip_csmat **out) {
    if (rows > 0xFFFFFFFFull || cols > 0xFFFFFFFFull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "triplet assembly: more than 2^32 rows or columns is not supported");
    if (in_idx_bytes == 8)
        return assemble<uint64_t>(rows, cols, n, (const uint64_t *)row_inds, (const uint64_t *)col_inds, data, storage, out_idx_bytes, out_iptr_bytes, out);
    return assemble<uint32_t>(rows, cols, n, (const uint32_t *)row_inds, (const uint32_t *)col_inds, data, storage, out_idx_bytes, out_iptr_bytes, out);
}

}  // namespace sprs_hip
