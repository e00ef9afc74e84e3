// C ABI of libsprs_hip.so (include/sprs_hip.h): handle lifecycle, raw device
// buffers, status / error plumbing.  Kernels live in spmv.hip and spgemm.hip.
#include <cstdarg>
#include <cstring>
#include <vector>

#include "common.hpp"
#include <map>

namespace sprs_hip {

static thread_local std::string tl_msg;
static thread_local int32_t tl_hip_code = 0;

void clear_error() {
    tl_msg.clear();
    tl_hip_code = 0;
}

void set_error(int32_t status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tl_msg = buf;
    if (status != SPRS_HIP_HIP_ERROR) tl_hip_code = 0;
}

int32_t fail_hip(hipError_t e, const char *what) {
    tl_hip_code = (int32_t)e;
    (void)hipGetLastError();   // clear the sticky error
    if (e == hipErrorOutOfMemory) {
        set_error(SPRS_HIP_OUT_OF_MEMORY, "out of device memory in %s", what);
        return SPRS_HIP_OUT_OF_MEMORY;
    }
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) {
        set_error(SPRS_HIP_NO_DEVICE, "no usable HIP device (%s) in %s", hipGetErrorString(e), what);
        tl_hip_code = (int32_t)e;
        return SPRS_HIP_NO_DEVICE;
    }
    char buf[400];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    tl_msg = buf;
    return SPRS_HIP_HIP_ERROR;
}

Options &options() {
    static Options o;
    return o;
}

// ---- pool of released result blocks ------------------------------------------------------------
// A product like config 5's is 2 x 26.5 GB.  The driver hands such blocks out in ~30 ms the first
// time, but takes SECONDS when the same amount has just been released (measured 2.0-3.4 s for the
// second A*A of a process, against 0.23 s of kernels: profiles/r01z_spgemm_v3_sweep.txt) — a loop
// that forms a product per iteration would spend 90 % of its time there.  Owned blocks of >= 1 MiB
// therefore go back to a per-device pool and the next result takes the best fit (<= 25 % slack).
// hipFree synchronises the device before a block can be reused; so does pool_free.
namespace {
struct Pool {
    std::mutex mu;
    std::multimap<std::pair<int, uint64_t>, void *> blocks;   // (device, bytes) -> block
    uint64_t cached = 0;
};
Pool &pool() {
    static Pool p;
    return p;
}
constexpr uint64_t POOL_MIN = 1ull << 20;      // from here on: best fit with <= 25 % slack
constexpr uint64_t POOL_SMALL = 256;            // below POOL_MIN: size classes, powers of two from this one
}  // namespace

uint64_t pool_trim() {
    Pool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    const uint64_t freed = p.cached;
    for (auto &kv : p.blocks) (void)hipFree(kv.second);
    p.blocks.clear();
    p.cached = 0;
    return freed;
}

uint64_t pool_cached_bytes() {
    Pool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    return p.cached;
}

hipError_t pool_alloc(void **out, uint64_t bytes, uint64_t *cap, int device) {
    if (bytes == 0) bytes = 8;       // hipMalloc(0) returns nullptr; kernels never see NULL
    // small blocks (task lists of a few entries, block sums of a scan, key arrays) are pooled in power-of-two classes from 256 B:
    // a product made ~15 of them, each a hipMalloc and a hipFree (which waits for the device) of its own
    if (bytes < POOL_MIN) {
        uint64_t cls = POOL_SMALL;
        while (cls < bytes) cls <<= 1;
        bytes = cls;
    }
    if (options().pool) {
        Pool &p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        auto it = p.blocks.lower_bound({device, bytes});
        if (it != p.blocks.end() && it->first.first == device && it->first.second <= bytes + bytes / 4) {
            *out = it->second;
            *cap = it->first.second;
            p.cached -= it->first.second;
            p.blocks.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory && pool_trim()) {
        (void)hipGetLastError();
        e = hipMalloc(out, bytes);
    }
    *cap = bytes;
    return e;
}

void pool_free(void *ptr, uint64_t cap, int device, bool stream_ordered) {
    if (!ptr) return;
    int current = -1;
    // a block of ANOTHER device than the calling thread's current one goes straight back to the driver:
    // the synchronisation below would wait on the wrong device
    if (options().pool && cap >= POOL_SMALL && hipGetDevice(&current) == hipSuccess && current == device) {
        Pool &p = pool();
        // the block may still be read by kernels in flight: same guarantee as hipFree — unless the caller vouches that all of
        // them, and every later use of a pooled block, are ordered by the null stream (the SpGEMM plan's temporaries: a product
        // dropped ~25 blocks, each behind its own device synchronisation)
        if (stream_ordered || hipDeviceSynchronize() == hipSuccess) {
            std::lock_guard<std::mutex> g(p.mu);
            if (p.cached + cap <= (uint64_t)options().pool_max_bytes) {
                p.blocks.insert({{device, cap}, ptr});
                p.cached += cap;
                return;
            }
        }
    }
    (void)hipFree(ptr);
}

void SpmvPlan::release() {
    auto drop = [](void *p) {
        if (p) (void)hipFree(p);
    };
    band_free(band);
    band = nullptr;
    drop(main.tile_row);
    drop(main.pos);
    if (main.owns) {
        drop(main.indptr);
        drop(main.indices);
        drop(main.data);
    }
    main = CsrPiece();
    for (auto &sl : slice) {
        drop(sl.tile_row);
        drop(sl.pos);
        if (sl.owns) {
            drop(sl.indptr);
            drop(sl.indices);
            drop(sl.data);
        }
        sl = CsrPiece();
    }
    drop(long_rows);
    drop(slab);
    drop(perm);
    perm = nullptr;
    long_rows = nullptr;
    slab = nullptr;
    for (auto &kv : scratch) {
        drop(kv.second.carry_main);
        drop(kv.second.carry_slices);
        drop(kv.second.partial);
        drop(kv.second.xp);
    }
    scratch.clear();
    n_long = 0;
    built = false;
    xcs = false;
}

int32_t alloc_csmat(sprs_hip_csmat **out, int32_t storage, uint64_t rows, uint64_t cols, uint64_t nnz,
                    int32_t iptr_bytes, int32_t idx_bytes) {
    auto *m = new sprs_hip_csmat();
    m->storage = storage;
    m->rows = rows;
    m->cols = cols;
    m->nnz = nnz;
    m->iptr_bytes = iptr_bytes;
    m->idx_bytes = idx_bytes;
    m->owns = true;
    hipError_t e = hipGetDevice(&m->device);
    if (e == hipSuccess) e = pool_alloc(&m->indptr, (m->outer() + 1) * (uint64_t)iptr_bytes, &m->cap_indptr, m->device);
    if (e == hipSuccess) e = pool_alloc(&m->indices, nnz * (uint64_t)idx_bytes, &m->cap_indices, m->device);
    if (e == hipSuccess) e = pool_alloc((void **)&m->data, nnz * sizeof(double), &m->cap_data, m->device);
    if (e != hipSuccess) {
        pool_free(m->indptr, m->cap_indptr, m->device);
        pool_free(m->indices, m->cap_indices, m->device);
        pool_free(m->data, m->cap_data, m->device);
        delete m;
        return fail_hip(e, "alloc_csmat");
    }
    *out = m;
    return SPRS_HIP_OK;
}

// utils::check_compressed_structure (sprs/src/sparse.rs:300-358) on host arrays.
template <typename I, typename P>
static int32_t check_structure(uint64_t inner, uint64_t outer, const P *indptr, const I *indices) {
    // Iptr / I must be able to represent the dimensions (sparse.rs:314-324)
    if ((uint64_t)(I)inner != inner) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Index type not large enough for this matrix");
    if ((uint64_t)(P)(outer + 1) != outer + 1) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Iptr type not large enough for this matrix");
    const uint64_t off = (uint64_t)indptr[0];
    for (uint64_t i = 0; i < outer; ++i)
        if (indptr[i + 1] < indptr[i]) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Unsorted indptr");
    if ((uint64_t)indptr[outer] > (UINT64_MAX >> 1)) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "An indptr value is larger than allowed");
    for (uint64_t r = 0; r < outer; ++r) {
        const uint64_t s = (uint64_t)indptr[r] - off, e = (uint64_t)indptr[r + 1] - off;
        for (uint64_t p = s + 1; p < e; ++p)
            if (indices[p] <= indices[p - 1]) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Indices are not sorted");
        if (e > s && (uint64_t)indices[e - 1] >= inner)
            SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Indice is larger than inner dimension");
    }
    return SPRS_HIP_OK;
}

static bool widths_ok(int32_t iptr_bytes, int32_t idx_bytes) {
    return (iptr_bytes == 4 || iptr_bytes == 8) && (idx_bytes == 4 || idx_bytes == 8);
}

static bool host_widths_ok(int32_t iptr_bytes, int32_t idx_bytes) {     // what a host array may hold: u16 too (indexing.rs:124-130)
    return (iptr_bytes == 2 || iptr_bytes == 4 || iptr_bytes == 8) && (idx_bytes == 2 || idx_bytes == 4 || idx_bytes == 8);
}

static uint64_t width_max(int32_t bytes) { return bytes >= 8 ? ~0ull : ((1ull << (8 * bytes)) - 1ull); }

int32_t inherit_declared_widths(sprs_hip_csmat *result, const sprs_hip_csmat *from) {
    result->decl_idx_bytes = from->decl_idx_bytes;
    result->decl_iptr_bytes = from->decl_iptr_bytes;
    // I::from_usize on the inner dimension / indices, Iptr::from_usize on the nnz (sparse.rs:314-324, smmp.rs:121, csmat.rs:1794)
    if (result->inner() && result->inner() - 1 > width_max(result->user_idx_bytes()))
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold the number of rows requested (required %llu)",
                  (unsigned long long)result->inner());
    if (result->nnz > width_max(result->user_iptr_bytes()))
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold the nnz of the result (%llu)", (unsigned long long)result->nnz);
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip

using namespace sprs_hip;

// a freshly made result takes the declared index widths of the operand it derives from; on overflow it is released
static int32_t finish_result(sprs_hip_csmat **res, const sprs_hip_csmat *from) {
    const int32_t st = inherit_declared_widths(*res, from);
    if (st != SPRS_HIP_OK) {
        const std::string keep = sprs_hip_last_error();
        sprs_hip_csmat_free(*res);
        *res = nullptr;
        set_error(st, "%s", keep.c_str());
    }
    return st;
}

extern "C" {

const char *sprs_hip_last_error(void) { return tl_msg.c_str(); }
int32_t sprs_hip_last_hip_code(void) { return tl_hip_code; }
const char *sprs_hip_version(void) { return "sprs_hip 0.1.0 gfx950 (twin of sprs 0.11.5 prod/smmp)"; }

int32_t sprs_hip_device_count(int32_t *count) {
    clear_error();
    if (!count) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(e, "hipGetDeviceCount");
    }
    *count = n;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_set_device(int32_t device) {
    clear_error();
    SPRS_TRY_HIP(hipSetDevice(device));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_malloc(void **dev_ptr, uint64_t bytes) {
    clear_error();
    if (!dev_ptr) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "dev_ptr is NULL");
    *dev_ptr = nullptr;
    SPRS_TRY_HIP(hipMalloc(dev_ptr, bytes ? bytes : 8));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_free(void *dev_ptr) {
    clear_error();
    if (dev_ptr) SPRS_TRY_HIP(hipFree(dev_ptr));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_memcpy_h2d(void *dev_dst, const void *host_src, uint64_t bytes) {
    clear_error();
    if (bytes) SPRS_TRY_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_memcpy_d2h(void *host_dst, const void *dev_src, uint64_t bytes) {
    clear_error();
    if (bytes) SPRS_TRY_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_memcpy_d2d(void *dev_dst, const void *dev_src, uint64_t bytes, void *stream) {
    clear_error();
    if (bytes) SPRS_TRY_HIP(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_memset(void *dev_dst, int32_t byte_value, uint64_t bytes, void *stream) {
    clear_error();
    if (bytes) SPRS_TRY_HIP(hipMemsetAsync(dev_dst, byte_value, bytes, (hipStream_t)stream));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_pool_trim(uint64_t *freed_bytes) {
    clear_error();
    const uint64_t f = pool_trim();
    if (freed_bytes) *freed_bytes = f;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_synchronize(void *stream) {
    clear_error();
    if (stream) SPRS_TRY_HIP(hipStreamSynchronize((hipStream_t)stream));
    else SPRS_TRY_HIP(hipDeviceSynchronize());
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_upload(sprs_hip_csmat **out, int32_t storage, uint64_t rows, uint64_t cols,
                              const void *indptr, int32_t iptr_bytes, const void *indices,
                              int32_t idx_bytes, const double *data, int32_t validate) {
    clear_error();
    if (!out || !indptr) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (storage != SPRS_HIP_CSR && storage != SPRS_HIP_CSC) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "bad storage tag %d", storage);
    if (!host_widths_ok(iptr_bytes, idx_bytes)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "index widths must be 2, 4 or 8 bytes");
    const uint64_t outer = storage == SPRS_HIP_CSR ? rows : cols;
    const uint64_t inner = storage == SPRS_HIP_CSR ? cols : rows;
    if (iptr_bytes == 2 || idx_bytes == 2) {
        // 2-byte index types (u16 / i16 in sprs): widened to 4 bytes for the device, the declared widths remembered
        const uint64_t first16 = iptr_bytes == 2 ? ((const uint16_t *)indptr)[0] : iptr_bytes == 4 ? ((const uint32_t *)indptr)[0] : ((const uint64_t *)indptr)[0];
        const uint64_t last16 = iptr_bytes == 2 ? ((const uint16_t *)indptr)[outer] : iptr_bytes == 4 ? ((const uint32_t *)indptr)[outer] : ((const uint64_t *)indptr)[outer];
        if (last16 < first16) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Unsorted indptr");
        const uint64_t nnz16 = last16 - first16;
        if (nnz16 && (!indices || !data)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL indices/data with nnz > 0");   // before the widening loop reads them
        if (inner && inner - 1 > width_max(idx_bytes)) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Index type not large enough for this matrix");
        // (a 2-byte indptr cannot hold a count above its own range: nothing to check for Iptr on the way in)
        std::vector<uint32_t> ip32, ix32;
        const void *ipp = indptr, *ixp = indices;
        int32_t ipb = iptr_bytes, ixb = idx_bytes;
        if (iptr_bytes == 2) {
            ip32.resize(outer + 1);
            for (uint64_t i = 0; i <= outer; ++i) ip32[i] = ((const uint16_t *)indptr)[i];
            ipp = ip32.data();
            ipb = 4;
        }
        if (idx_bytes == 2) {
            ix32.resize(nnz16 ? nnz16 : 1);
            const uint16_t *src = (const uint16_t *)indices;   // points at the element addressed by indptr[0]
            for (uint64_t i = 0; i < nnz16; ++i) ix32[i] = src[i];
            ixp = nnz16 ? ix32.data() : indices;
            ixb = 4;
        }
        SPRS_TRY(sprs_hip_csmat_upload(out, storage, rows, cols, ipp, ipb, ixp, ixb, data, validate));
        (*out)->decl_iptr_bytes = iptr_bytes;
        (*out)->decl_idx_bytes = idx_bytes;
        return SPRS_HIP_OK;
    }
    auto ip_at = [&](uint64_t i) -> uint64_t {
        return iptr_bytes == 8 ? ((const uint64_t *)indptr)[i] : ((const uint32_t *)indptr)[i];
    };
    const uint64_t first = ip_at(0), last = ip_at(outer);
    if (last < first) SPRS_FAIL(SPRS_HIP_BAD_STRUCTURE, "Unsorted indptr");
    const uint64_t nnz = last - first;
    if (nnz && (!indices || !data)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL indices/data with nnz > 0");
    if (validate) {
        int32_t st;
        if (iptr_bytes == 8 && idx_bytes == 8) st = check_structure(inner, outer, (const uint64_t *)indptr, (const uint64_t *)indices);
        else if (iptr_bytes == 8) st = check_structure(inner, outer, (const uint64_t *)indptr, (const uint32_t *)indices);
        else if (idx_bytes == 8) st = check_structure(inner, outer, (const uint32_t *)indptr, (const uint64_t *)indices);
        else st = check_structure(inner, outer, (const uint32_t *)indptr, (const uint32_t *)indices);
        SPRS_TRY(st);
    }
    sprs_hip_csmat *m = nullptr;
    SPRS_TRY(alloc_csmat(&m, storage, rows, cols, nnz, iptr_bytes, idx_bytes));
    hipError_t e = hipSuccess;
    if (first == 0) {
        e = hipMemcpy(m->indptr, indptr, (outer + 1) * (uint64_t)iptr_bytes, hipMemcpyHostToDevice);
    } else {
        // to_proper (sprs/src/sparse/indptr.rs:206-214): rebase on the way in
        std::vector<uint8_t> tmp((outer + 1) * (size_t)iptr_bytes);
        for (uint64_t i = 0; i <= outer; ++i) {
            if (iptr_bytes == 8) ((uint64_t *)tmp.data())[i] = ip_at(i) - first;
            else ((uint32_t *)tmp.data())[i] = (uint32_t)(ip_at(i) - first);
        }
        e = hipMemcpy(m->indptr, tmp.data(), tmp.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && nnz) e = hipMemcpy(m->indices, indices, nnz * (uint64_t)idx_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(m->data, data, nnz * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        sprs_hip_csmat_free(m);
        return fail_hip(e, "csmat_upload");
    }
    *out = m;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_wrap_device(sprs_hip_csmat **out, int32_t storage, uint64_t rows,
                                   uint64_t cols, uint64_t nnz, const void *dev_indptr,
                                   int32_t iptr_bytes, const void *dev_indices, int32_t idx_bytes,
                                   const double *dev_data) {
    clear_error();
    if (!out || !dev_indptr) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (storage != SPRS_HIP_CSR && storage != SPRS_HIP_CSC) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "bad storage tag %d", storage);
    if (!widths_ok(iptr_bytes, idx_bytes)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "index widths must be 4 or 8 bytes");
    if (nnz && (!dev_indices || !dev_data)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL indices/data with nnz > 0");
    if (((uintptr_t)dev_indptr | (uintptr_t)dev_indices | (uintptr_t)dev_data) & 15)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "device buffers must be 16-byte aligned");
    auto *m = new sprs_hip_csmat();
    m->storage = storage;
    m->rows = rows;
    m->cols = cols;
    m->nnz = nnz;
    m->iptr_bytes = iptr_bytes;
    m->idx_bytes = idx_bytes;
    m->indptr = const_cast<void *>(dev_indptr);
    m->indices = const_cast<void *>(dev_indices);
    m->data = const_cast<double *>(dev_data);
    m->owns = false;
    hipError_t e = hipGetDevice(&m->device);
    if (e != hipSuccess) {
        delete m;
        return fail_hip(e, "hipGetDevice");
    }
    *out = m;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_info(const sprs_hip_csmat *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz,
                            int32_t *iptr_bytes, int32_t *idx_bytes, int32_t *storage) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (nnz) *nnz = m->nnz;
    if (iptr_bytes) *iptr_bytes = m->user_iptr_bytes();       // the widths the caller declared (2-byte arrays live widened on the device)
    if (idx_bytes) *idx_bytes = m->user_idx_bytes();
    if (storage) *storage = m->storage;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_device_ptrs(const sprs_hip_csmat *m, const void **indptr,
                                   const void **indices, const double **data) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    if (indptr) *indptr = m->indptr;
    if (indices) *indices = m->indices;
    if (data) *data = m->data;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_download(const sprs_hip_csmat *m, void *indptr, void *indices, double *data) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    auto fetch = [&](void *dst, const void *src, uint64_t count, int32_t dev_bytes, int32_t user_bytes) -> int32_t {
        if (user_bytes == dev_bytes) {
            SPRS_TRY_HIP(hipMemcpy(dst, src, count * (uint64_t)dev_bytes, hipMemcpyDeviceToHost));
            return SPRS_HIP_OK;
        }
        std::vector<uint32_t> wide(count ? count : 1);          // declared 2 bytes, device 4: narrow (every value was range-checked when made)
        SPRS_TRY_HIP(hipMemcpy(wide.data(), src, count * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < count; ++i) {
            if (wide[i] > 0xFFFFu) SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold %u", wide[i]);
            ((uint16_t *)dst)[i] = (uint16_t)wide[i];
        }
        return SPRS_HIP_OK;
    };
    if (indptr) SPRS_TRY(fetch(indptr, m->indptr, m->outer() + 1, m->iptr_bytes, m->user_iptr_bytes()));
    if (indices && m->nnz) SPRS_TRY(fetch(indices, m->indices, m->nnz, m->idx_bytes, m->user_idx_bytes()));
    if (data && m->nnz) SPRS_TRY_HIP(hipMemcpy(data, m->data, m->nnz * sizeof(double), hipMemcpyDeviceToHost));
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_download_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end,
                                      void *indptr_out, void *indices_out, double *data_out,
                                      uint64_t *nnz_out) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    // Range checks of slice_outer (sprs/src/sparse/slicing.rs:65-89 -> range.rs)
    if (start > end || end > m->outer()) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "slice_outer range out of bounds");
    uint64_t lo = 0, hi = 0;
    const uint64_t pb = (uint64_t)m->iptr_bytes;
    if (pb == 8) {
        SPRS_TRY_HIP(hipMemcpy(&lo, (const uint8_t *)m->indptr + start * pb, 8, hipMemcpyDeviceToHost));
        SPRS_TRY_HIP(hipMemcpy(&hi, (const uint8_t *)m->indptr + end * pb, 8, hipMemcpyDeviceToHost));
    } else {
        uint32_t a = 0, b = 0;
        SPRS_TRY_HIP(hipMemcpy(&a, (const uint8_t *)m->indptr + start * pb, 4, hipMemcpyDeviceToHost));
        SPRS_TRY_HIP(hipMemcpy(&b, (const uint8_t *)m->indptr + end * pb, 4, hipMemcpyDeviceToHost));
        lo = a;
        hi = b;
    }
    if (nnz_out) *nnz_out = hi - lo;
    auto fetch = [&](void *dst, const void *src, uint64_t count, int32_t dev_bytes, int32_t user_bytes) -> int32_t {
        if (user_bytes == dev_bytes) {
            SPRS_TRY_HIP(hipMemcpy(dst, src, count * (uint64_t)dev_bytes, hipMemcpyDeviceToHost));
            return SPRS_HIP_OK;
        }
        std::vector<uint32_t> wide(count ? count : 1);          // declared 2 bytes, device 4
        SPRS_TRY_HIP(hipMemcpy(wide.data(), src, count * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < count; ++i) {
            if (wide[i] > 0xFFFFu) SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold %u", wide[i]);
            ((uint16_t *)dst)[i] = (uint16_t)wide[i];
        }
        return SPRS_HIP_OK;
    };
    if (indptr_out) SPRS_TRY(fetch(indptr_out, (const uint8_t *)m->indptr + start * pb, end - start + 1, m->iptr_bytes, m->user_iptr_bytes()));
    if (indices_out && hi > lo)
        SPRS_TRY(fetch(indices_out, (const uint8_t *)m->indices + lo * (uint64_t)m->idx_bytes, hi - lo, m->idx_bytes, m->user_idx_bytes()));
    if (data_out && hi > lo) SPRS_TRY_HIP(hipMemcpy(data_out, m->data + lo, (hi - lo) * sizeof(double), hipMemcpyDeviceToHost));
    return SPRS_HIP_OK;
}

// Everything a handle derives from its arrays: the SpMV / SpMM / Gauss-Seidel plans (they copy values), the copy in the other
// storage order and the transpose view kept for dense . sparse products.  Called whenever the values (or the arrays) change:
// refresh, free and BOTH numeric SpGEMM entries, which rewrite C's values in place (a stale CSC copy of C
// would otherwise serve later products).
static void invalidate_caches(sprs_hip_csmat *m) {
    m->plan.release();
    m->mm.release();
    m->gs.release();
    if (m->t_view) (void)sprs_hip_csmat_free(m->t_view);
    m->t_view = nullptr;
    if (m->as_other) (void)sprs_hip_csmat_free(m->as_other);
    m->as_other = nullptr;
}

int32_t sprs_hip_csmat_refresh(sprs_hip_csmat *m) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    SPRS_TRY_HIP(hipDeviceSynchronize());     // nothing in flight may still read the plan copies
    std::lock_guard<std::recursive_mutex> lock(m->mu);
    invalidate_caches(m);
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_spmv_plan_info(const sprs_hip_csmat *m, int32_t *kind, uint64_t *plan_bytes) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    auto *mm = const_cast<sprs_hip_csmat *>(m);
    std::lock_guard<std::recursive_mutex> lock(mm->mu);
    const SpmvPlan &pl = m->plan;
    int32_t k = 0;
    uint64_t bytes = 0;
    if (pl.built) {
        if (pl.band) {
            k = 3;
            bytes = band_plan_bytes(pl.band);
        } else if (pl.xcs) {
            k = 2;
            bytes = pl.main.nnz * (8 + (uint64_t)pl.idx_bytes);
            for (const auto &sl : pl.slice) bytes += sl.nnz * (8 + (uint64_t)pl.idx_bytes) + (sl.rows + 1) * 8;
        } else {
            k = 1;
            bytes = (pl.main.ntiles + 1) * 8;
        }
    }
    if (kind) *kind = k;
    if (plan_bytes) *plan_bytes = bytes;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_transpose_view(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    clear_error();
    if (!m || !out) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    auto *t = new sprs_hip_csmat();
    t->storage = m->storage == SPRS_HIP_CSR ? SPRS_HIP_CSC : SPRS_HIP_CSR;
    t->rows = m->cols;
    t->cols = m->rows;
    t->nnz = m->nnz;
    t->iptr_bytes = m->iptr_bytes;
    t->idx_bytes = m->idx_bytes;
    t->decl_iptr_bytes = m->decl_iptr_bytes;
    t->decl_idx_bytes = m->decl_idx_bytes;
    t->indptr = m->indptr;
    t->indices = m->indices;
    t->data = m->data;
    t->owns = false;
    t->device = m->device;
    *out = t;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_csmat_free(sprs_hip_csmat *m) {
    clear_error();
    if (!m) return SPRS_HIP_OK;
    // a kernel launched through a cached copy (as_other, t_view) or a plan copy may still be reading it: a handle that holds
    // any of them waits for the device before they go (refresh does the same; a bare handle frees at once)
    if (m->t_view || m->as_other || m->plan.built) (void)hipDeviceSynchronize();
    invalidate_caches(m);
    if (m->owns) {
        pool_free(m->indptr, m->cap_indptr, m->device);
        pool_free(m->indices, m->cap_indices, m->device);
        pool_free(m->data, m->cap_data, m->device);
    }
    delete m;
    return SPRS_HIP_OK;
}

int32_t sprs_hip_spmv_f64(const sprs_hip_csmat *a, const double *x_dev, uint64_t x_len,
                          double *y_dev, uint64_t y_len, int32_t accumulate, void *stream) {
    clear_error();
    if (!a) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    // prod.rs:114-118: dimension check first, then storage
    if (a->cols != x_len || a->rows != y_len) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (a->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    if ((x_len && !x_dev) || (y_len && !y_dev)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL vector");
    // &mut res_vec cannot alias &in_vec in the reference; here the kernels read x while they write y
    if (x_len && y_len && x_dev < y_dev + y_len && y_dev < x_dev + x_len)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "the result vector overlaps the input vector");
    return spmv_f64(const_cast<sprs_hip_csmat *>(a), x_dev, y_dev, accumulate != 0, (hipStream_t)stream);
}

int32_t sprs_hip_spmv_f64_host(uint64_t rows, uint64_t cols, const void *indptr,
                               int32_t iptr_bytes, const void *indices, int32_t idx_bytes,
                               const double *data, const double *x, uint64_t x_len, double *y,
                               uint64_t y_len, int32_t accumulate) {
    clear_error();
    if (cols != x_len || rows != y_len) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    sprs_hip_csmat *m = nullptr;
    SPRS_TRY(sprs_hip_csmat_upload(&m, SPRS_HIP_CSR, rows, cols, indptr, iptr_bytes, indices, idx_bytes, data, 0));
    double *dx = nullptr, *dy = nullptr;
    int32_t st = sprs_hip_malloc((void **)&dx, x_len * 8);
    if (st == SPRS_HIP_OK) st = sprs_hip_malloc((void **)&dy, y_len * 8);
    if (st == SPRS_HIP_OK) st = sprs_hip_memcpy_h2d(dx, x, x_len * 8);
    if (st == SPRS_HIP_OK && accumulate) st = sprs_hip_memcpy_h2d(dy, y, y_len * 8);
    // one multiply, then the handle goes: nothing would amortise a re-laid-out copy of the matrix (the banded plan takes
    // ~0.1 s to build on R-MAT 10M for a 1 ms SpMV), so this entry multiplies on the plain nnz-tiled plan
    m->one_shot = true;
    if (st == SPRS_HIP_OK) st = sprs_hip_spmv_f64(m, dx, x_len, dy, y_len, accumulate, nullptr);
    if (st == SPRS_HIP_OK) st = sprs_hip_synchronize(nullptr);
    if (st == SPRS_HIP_OK) st = sprs_hip_memcpy_d2h(y, dy, y_len * 8);
    std::string keep = tl_msg;
    int32_t keep_code = tl_hip_code;
    (void)hipFree(dx);
    (void)hipFree(dy);
    sprs_hip_csmat_free(m);
    if (st != SPRS_HIP_OK) {
        tl_msg = keep;
        tl_hip_code = keep_code;
    }
    return st;
}

int32_t sprs_hip_spmm_rowmaj_f64(const sprs_hip_csmat *a, const double *rhs_dev, uint64_t rhs_rows, uint64_t k,
                                 uint64_t ld_rhs, double *out_dev, uint64_t out_rows, uint64_t ld_out,
                                 int32_t accumulate, void *stream) {
    clear_error();
    if (!a) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    // prod.rs:199-202: three dimension asserts, then storage
    if (a->cols != rhs_rows || a->rows != out_rows) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (a->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    if (ld_rhs < k || ld_out < k) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "leading dimension smaller than the column count");
    if (k && ((rhs_rows && !rhs_dev) || (out_rows && !out_dev))) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL matrix");
    return spmm_rowmaj_f64(const_cast<sprs_hip_csmat *>(a), rhs_dev, k, ld_rhs, out_dev, ld_out, accumulate != 0,
                           (hipStream_t)stream);
}

static int32_t spgemm_contract(const sprs_hip_csmat *a, const sprs_hip_csmat *b) {
    if (a->cols != b->rows) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");   // smmp.rs:207
    if (a->storage != SPRS_HIP_CSR || b->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    if (a->user_iptr_bytes() != b->user_iptr_bytes() || a->user_idx_bytes() != b->user_idx_bytes())
        SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "operands must share index types (smmp.rs:196-199)");
    return SPRS_HIP_OK;
}

static int32_t numeric_target_ok(const sprs_hip_csmat *a, const sprs_hip_csmat *b, const sprs_hip_csmat *c) {
    // smmp.rs:161-166: the asserts of numeric() on the shape of c
    if (c->rows != a->rows || c->cols != b->cols) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (c->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    if (c->user_iptr_bytes() != a->user_iptr_bytes() || c->user_idx_bytes() != a->user_idx_bytes())
        SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "C must share the operands' index types");
    return SPRS_HIP_OK;
}

int32_t sprs_hip_spgemm_symbolic(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c) {
    clear_error();
    if (!a || !b || !c) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *c = nullptr;
    SPRS_TRY(spgemm_contract(a, b));
    SPRS_TRY(spgemm_symbolic(a, b, c));
    return finish_result(c, a);
}

int32_t sprs_hip_spgemm_numeric(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat *c) {
    clear_error();
    if (!a || !b || !c) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    SPRS_TRY(spgemm_contract(a, b));
    SPRS_TRY(numeric_target_ok(a, b, c));
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    SPRS_TRY_HIP(hipDeviceSynchronize());   // nothing in flight may still read the copies that go away
    invalidate_caches(c);                   // values are about to change: plans copy them, and so does the copy in the other storage order
    return spgemm_numeric(a, b, c);
}

int32_t sprs_hip_spgemm_plan_create(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_spgemm_plan **plan) {
    clear_error();
    if (!a || !b || !plan) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *plan = nullptr;
    SPRS_TRY(spgemm_contract(a, b));
    return spgemm_plan_create(a, b, plan);
}

int32_t sprs_hip_spgemm_plan_nnz(const sprs_hip_spgemm_plan *plan, uint64_t *nnz) {
    clear_error();
    if (!plan || !nnz) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *nnz = spgemm_plan_nnz(plan);
    return SPRS_HIP_OK;
}

int32_t sprs_hip_spgemm_plan_structure(sprs_hip_spgemm_plan *plan, const sprs_hip_csmat *a, const sprs_hip_csmat *b,
                                       sprs_hip_csmat **c_structure) {
    clear_error();
    if (!plan || !a || !b || !c_structure) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *c_structure = nullptr;
    SPRS_TRY(spgemm_plan_structure(plan, a, b, c_structure, false));
    return finish_result(c_structure, a);
}

int32_t sprs_hip_spgemm_plan_product(sprs_hip_spgemm_plan *plan, const sprs_hip_csmat *a, const sprs_hip_csmat *b,
                                     sprs_hip_csmat **c) {
    clear_error();
    if (!plan || !a || !b || !c) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *c = nullptr;
    SPRS_TRY(spgemm_plan_structure(plan, a, b, c, true));
    return finish_result(c, a);
}

int32_t sprs_hip_spgemm_plan_numeric(sprs_hip_spgemm_plan *plan, const sprs_hip_csmat *a, const sprs_hip_csmat *b,
                                     sprs_hip_csmat *c) {
    clear_error();
    if (!plan || !a || !b || !c) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    SPRS_TRY(numeric_target_ok(a, b, c));
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    SPRS_TRY_HIP(hipDeviceSynchronize());   // nothing in flight may still read the copies that go away
    invalidate_caches(c);                   // values are about to change: plans copy them, and so does the copy in the other storage order
    return spgemm_plan_numeric(plan, a, b, c);
}

int32_t sprs_hip_spgemm_plan_free(sprs_hip_spgemm_plan *plan) {
    clear_error();
    spgemm_plan_free(plan);
    return SPRS_HIP_OK;
}

int32_t sprs_hip_bicgstab_f64(sprs_hip_csmat *a, const double *x0_dev, const double *b_dev, uint64_t n, double tol,
                              uint64_t max_iter, double soft_restart_threshold, double *x_dev,
                              sprs_hip_bicgstab_info *info, void *stream) {
    clear_error();
    if (!a) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    if (a->rows != a->cols || a->rows != n) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (n && (!x0_dev || !b_dev || !x_dev)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL vector");
    if (x_dev == x0_dev || x_dev == b_dev) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "x_dev may not alias x0_dev / b_dev");
    if (n == 0) {
        if (info) *info = sprs_hip_bicgstab_info{0, 0, 0, 0.0, 0.0, 1};
        return SPRS_HIP_OK;
    }
    return bicgstab_f64(a, x0_dev, b_dev, n, tol, max_iter, soft_restart_threshold, x_dev, info, (hipStream_t)stream);
}

int32_t sprs_hip_gauss_seidel_f64(sprs_hip_csmat *a, double *x_dev, const double *rhs_dev, uint64_t n, uint64_t max_iter,
                                  double eps, sprs_hip_gauss_seidel_info *info, void *stream) {
    clear_error();
    if (!a) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    if (a->rows != a->cols || a->rows != n) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");   // heat.rs:109-110
    if (a->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Gauss-Seidel sweeps the rows of a CSR matrix");
    if (n && (!x_dev || !rhs_dev)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL vector");
    if (x_dev && x_dev == rhs_dev) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "x_dev may not alias rhs_dev");
    return gauss_seidel_f64(a, x_dev, rhs_dev, n, max_iter, eps, info, (hipStream_t)stream);
}

int32_t sprs_hip_spgemm_f64(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c) {
    clear_error();
    if (!a || !b || !c) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *c = nullptr;
    if (a->cols != b->rows) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");   // smmp.rs:207
    if (a->storage != SPRS_HIP_CSR || b->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    if (a->user_iptr_bytes() != b->user_iptr_bytes() || a->user_idx_bytes() != b->user_idx_bytes())
        SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "operands must share index types (smmp.rs:196-199)");
    SPRS_TRY(spgemm_f64(a, b, c));
    return finish_result(c, a);
}

// to_other_storage with the index-type check of the reference on the DECLARED widths (csmat.rs:1794-1797)
static int32_t convert_checked(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    if (m->rows > width_max(m->user_idx_bytes()))
        SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type is not large enough to hold the number of rows requested (required %llu)",
                  (unsigned long long)m->rows);
    SPRS_TRY(to_other_storage(m, out));
    return finish_result(out, m);
}

int32_t sprs_hip_csmat_to_other_storage(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    clear_error();
    if (!m || !out) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *out = nullptr;
    return convert_checked(m, out);
}

// `&lhs * &rhs` for two sparse matrices: csmat_mul_csmat (csmat.rs:1895-1949) — the storage dispatch around
// smmp::mul_csr_csr; the result has the storage of the lhs.
int32_t sprs_hip_csmat_mul_csmat(const sprs_hip_csmat *lhs, const sprs_hip_csmat *rhs, sprs_hip_csmat **out) {
    clear_error();
    if (!lhs || !rhs || !out) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (lhs->cols != rhs->rows) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (lhs->user_iptr_bytes() != rhs->user_iptr_bytes() || lhs->user_idx_bytes() != rhs->user_idx_bytes())
        SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "operands must share index types (smmp.rs:196-199)");
    auto view_t = [](const sprs_hip_csmat *m, sprs_hip_csmat &t) {      // transpose_view (csmat.rs:982-991): free, flips the tag
        t.storage = m->storage == SPRS_HIP_CSR ? SPRS_HIP_CSC : SPRS_HIP_CSR;
        t.rows = m->cols;
        t.cols = m->rows;
        t.nnz = m->nnz;
        t.iptr_bytes = m->iptr_bytes;
        t.idx_bytes = m->idx_bytes;
        t.decl_iptr_bytes = m->decl_iptr_bytes;
        t.decl_idx_bytes = m->decl_idx_bytes;
        t.indptr = m->indptr;
        t.indices = m->indices;
        t.data = m->data;
        t.owns = false;
        t.device = m->device;
    };
    struct Owned {
        sprs_hip_csmat *h = nullptr;
        ~Owned() {
            if (h) sprs_hip_csmat_free(h);
        }
    };
    const bool l_csr = lhs->storage == SPRS_HIP_CSR, r_csr = rhs->storage == SPRS_HIP_CSR;
    if (l_csr && r_csr) {                                         // (CSR, CSR)
        SPRS_TRY(spgemm_f64(lhs, rhs, out));
        return finish_result(out, lhs);
    }
    if (l_csr) {                                                  // (CSR, CSC): rhs.to_other_storage()
        Owned conv;
        SPRS_TRY(convert_checked(rhs, &conv.h));
        SPRS_TRY(spgemm_f64(lhs, conv.h, out));
        return finish_result(out, lhs);
    }
    // lhs is CSC: (rhs^T * lhs^T)^T on the transpose views, which are CSR; transpose_into flips the result back
    Owned conv;
    const sprs_hip_csmat *r = rhs;
    if (r_csr) {                                                  // (CSC, CSR): rhs.to_other_storage() first
        SPRS_TRY(convert_checked(rhs, &conv.h));
        r = conv.h;
    }
    sprs_hip_csmat rt, lt;
    view_t(r, rt);
    view_t(lhs, lt);
    SPRS_TRY(spgemm_f64(&rt, &lt, out));
    sprs_hip_csmat *c = *out;                                     // transpose_into: same buffers, other tag
    c->storage = SPRS_HIP_CSC;
    std::swap(c->rows, c->cols);
    return finish_result(out, lhs);
}

int32_t sprs_hip_csmat_slice_outer(const sprs_hip_csmat *m, uint64_t start, uint64_t end, sprs_hip_csmat **out) {
    clear_error();
    if (!m || !out) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (start > end || end > m->outer()) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "slice_outer range out of bounds");
    SPRS_TRY(slice_outer(m, start, end, out));
    return finish_result(out, m);
}

int32_t sprs_hip_dist_unique_id(void *id_128_bytes) {
    clear_error();
    if (!id_128_bytes) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    return dist_unique_id(id_128_bytes);
}

int32_t sprs_hip_dist_create(sprs_hip_dist **d, const void *id_128_bytes, int32_t world, int32_t rank, uint64_t rows,
                             uint64_t cols, const uint64_t *row_starts, const sprs_hip_csmat *local_block, int32_t nsub) {
    clear_error();
    if (!d || !row_starts || !local_block) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");   // (no id with world > 1: a handle for the peer route only)
    *d = nullptr;
    return dist_create(d, id_128_bytes, world, rank, rows, cols, row_starts, local_block, nsub);
}

int32_t sprs_hip_dist_spmv_f64(sprs_hip_dist *d, const double *x_dev, uint64_t x_len, double *y_dev, uint64_t y_len, void *stream) {
    clear_error();
    if (!d) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    if (dist_cols(d) != x_len || dist_rows(d) != y_len) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if ((x_len && !x_dev) || (y_len && !y_dev)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL vector");
    return dist_spmv(d, x_dev, y_dev, (hipStream_t)stream);
}

// ---- products with dense operands: the storage dispatch of the reference below the ABI -----------------------------------------
// The CSR form of a handle: itself, or — for a CSC handle — its to_other_storage() copy (csmat.rs:1405-1426), made once and kept
// IN THE HANDLE (as_other): every later product of the same CSC matrix reuses it, from any host language.  The reference walks a
// CSC matrix column by column and scatters (prod.rs:74-99, 219-270): every result element receives its products by ascending
// column index, which is the order of the CSR kernels on the converted matrix.
static int32_t other_form(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    auto *mm = const_cast<sprs_hip_csmat *>(m);
    std::lock_guard<std::recursive_mutex> lock(mm->mu);
    if (!mm->as_other) {
        SPRS_TRY(to_other_storage(m, &mm->as_other));
        // the plan policy (prepare flag, multiplies so far) is the OWNER's: a copy rebuilt after a refresh keeps it
        mm->as_other->prepared = mm->prepared;
        mm->as_other->spmv_calls = mm->spmv_calls;
    }
    *out = mm->as_other;
    return SPRS_HIP_OK;
}
static int32_t csr_form(const sprs_hip_csmat *m, sprs_hip_csmat **out) {
    if (m->storage == SPRS_HIP_CSR) {
        *out = const_cast<sprs_hip_csmat *>(m);
        return SPRS_HIP_OK;
    }
    return other_form(m, out);
}

int32_t sprs_hip_csmat_prepare(sprs_hip_csmat *m, void *stream) {
    clear_error();
    if (!m) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    sprs_hip_csmat *csr = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lock(m->mu);
        m->prepared = true;                    // kept on the handle the caller holds: survives a refresh of a CSC handle's CSR copy
    }
    SPRS_TRY(csr_form(m, &csr));               // a CSC handle multiplies through its CSR copy: that is the handle to prepare
    return spmv_prepare(csr, (hipStream_t)stream);
}

static int32_t vec_args_ok(const double *x_dev, uint64_t x_len, double *y_dev, uint64_t y_len) {
    if ((x_len && !x_dev) || (y_len && !y_dev)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL vector");
    if (x_len && y_len && x_dev < y_dev + y_len && y_dev < x_dev + x_len)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "the result vector overlaps the input vector");
    return SPRS_HIP_OK;
}

int32_t sprs_hip_mul_acc_mat_vec_csc_f64(const sprs_hip_csmat *a, const double *x_dev, uint64_t x_len, double *y_dev,
                                         uint64_t y_len, void *stream) {
    clear_error();
    if (!a) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    // prod.rs:88-92: dimensions first, then storage
    if (a->cols != x_len || a->rows != y_len) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (a->storage != SPRS_HIP_CSC) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    SPRS_TRY(vec_args_ok(x_dev, x_len, y_dev, y_len));
    sprs_hip_csmat *csr = nullptr;
    SPRS_TRY(csr_form(a, &csr));
    const int32_t st = spmv_f64(csr, x_dev, y_dev, true, (hipStream_t)stream);
    if (csr != a) const_cast<sprs_hip_csmat *>(a)->spmv_calls = csr->spmv_calls;
    return st;
}

int32_t sprs_hip_csmat_mul_vec_f64(const sprs_hip_csmat *a, const double *x_dev, uint64_t x_len, double *y_dev, uint64_t y_len,
                                   void *stream) {
    clear_error();
    if (!a) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    // csmat.rs:2119-2160 -> csr_/csc_mulacc_dense_colmaj with one column (prod.rs:284-289 / 257-259: "Dimension mismatch")
    if (a->cols != x_len || a->rows != y_len) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    SPRS_TRY(vec_args_ok(x_dev, x_len, y_dev, y_len));
    sprs_hip_csmat *csr = nullptr;
    SPRS_TRY(csr_form(a, &csr));
    const int32_t st = spmv_f64(csr, x_dev, y_dev, false, (hipStream_t)stream);
    if (csr != a) const_cast<sprs_hip_csmat *>(a)->spmv_calls = csr->spmv_calls;
    return st;
}

int32_t sprs_hip_csmat_mulacc_dense_f64(const sprs_hip_csmat *lhs, const double *rhs_dev, uint64_t rhs_rows, uint64_t k,
                                        int32_t rhs_layout, uint64_t ld_rhs, double *out_dev, uint64_t out_rows,
                                        int32_t out_layout, uint64_t ld_out, int32_t accumulate, void *stream) {
    clear_error();
    if (!lhs) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    if ((rhs_layout != SPRS_HIP_ROW_MAJOR && rhs_layout != SPRS_HIP_COL_MAJOR) || (out_layout != SPRS_HIP_ROW_MAJOR && out_layout != SPRS_HIP_COL_MAJOR))
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "bad layout tag");
    // prod.rs:199-201, 228-230, 257-259, 284-289: the three dimension asserts of the four dense kernels (the column counts of rhs
    // and out are one argument here)
    if (lhs->cols != rhs_rows || lhs->rows != out_rows) SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    if (ld_rhs < (rhs_layout == SPRS_HIP_ROW_MAJOR ? k : rhs_rows) || ld_out < (out_layout == SPRS_HIP_ROW_MAJOR ? k : out_rows))
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "leading dimension smaller than the extent it strides over");
    if (k && ((rhs_rows && !rhs_dev) || (out_rows && !out_dev))) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL matrix");
    if (k && rhs_rows && out_rows) {   // the two operands must not share an ELEMENT (as vec_args_ok for the vectors)
        const uint64_t span_r = (rhs_layout == SPRS_HIP_ROW_MAJOR ? rhs_rows - 1 : k - 1) * ld_rhs + (rhs_layout == SPRS_HIP_ROW_MAJOR ? k : rhs_rows);
        const uint64_t span_o = (out_layout == SPRS_HIP_ROW_MAJOR ? out_rows - 1 : k - 1) * ld_out + (out_layout == SPRS_HIP_ROW_MAJOR ? k : out_rows);
        if (rhs_dev < out_dev + span_o && out_dev < rhs_dev + span_r) {
            // the strided spans meet.  Two views of ONE buffer with the same layout and pitch — rhs = buf[:, 0:k], out = buf[:, k:2k],
            // or the same as column blocks; ndarray views the reference accepts — are still element-disjoint when the lines of one
            // fall into the gaps of the other: the offset between them, modulo the pitch, leaves both line extents apart.
            bool disjoint = false;
            if (rhs_layout == out_layout && ld_rhs == ld_out) {
                const uint64_t w_r = rhs_layout == SPRS_HIP_ROW_MAJOR ? k : rhs_rows, w_o = out_layout == SPRS_HIP_ROW_MAJOR ? k : out_rows;
                const uint64_t d = rhs_dev <= out_dev ? (uint64_t)(out_dev - rhs_dev) % ld_rhs : (ld_rhs - (uint64_t)(rhs_dev - out_dev) % ld_rhs) % ld_rhs;
                // inside one pitch: rhs lines occupy [0, w_r), out lines [d, d + w_o)
                disjoint = d >= w_r && d + w_o <= ld_rhs;
            }
            if (!disjoint) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "the result matrix overlaps the right-hand side");
        }
    }
    sprs_hip_csmat *csr = nullptr;
    SPRS_TRY(csr_form(lhs, &csr));
    hipStream_t st = (hipStream_t)stream;
    if (rhs_layout == SPRS_HIP_COL_MAJOR && out_layout == SPRS_HIP_COL_MAJOR && k < 8) {
        // the shape `&CsMat * &Array2` gives csr_mulacc_dense_colmaj (fewer than 8 columns, csmat.rs:2002-2016): column by column
        // through the SpMV — a column of a column-major rhs is a separate line per entry, so the SpMM kernel would issue k gathers
        // per entry where it issues one for a row-major rhs (k x 5.5 ms against k x 1.04 ms of the banded SpMV)
        for (uint64_t j = 0; j < k; ++j) SPRS_TRY(spmv_f64(csr, rhs_dev + j * ld_rhs, out_dev + j * ld_out, accumulate != 0, st));
        return SPRS_HIP_OK;
    }
    const uint64_t rs_r = rhs_layout == SPRS_HIP_ROW_MAJOR ? ld_rhs : 1, cs_r = rhs_layout == SPRS_HIP_ROW_MAJOR ? 1 : ld_rhs;
    const uint64_t rs_o = out_layout == SPRS_HIP_ROW_MAJOR ? ld_out : 1, cs_o = out_layout == SPRS_HIP_ROW_MAJOR ? 1 : ld_out;
    return spmm_strided_f64(csr, rhs_dev, k, rs_r, cs_r, out_dev, rs_o, cs_o, accumulate != 0, st);
}

int32_t sprs_hip_csmat_mul_dense_f64(const sprs_hip_csmat *lhs, const double *rhs_dev, uint64_t rhs_rows, uint64_t k,
                                     int32_t rhs_layout, uint64_t ld_rhs, double *out_dev, int32_t *out_layout, void *stream) {
    clear_error();
    if (!lhs || !out_layout) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    // csmat.rs:2002-2045: Array::zeros((rows, cols)) for >= 8 columns, Array::zeros((rows, cols).f()) below — in both storages
    const int32_t lay = k >= 8 ? SPRS_HIP_ROW_MAJOR : SPRS_HIP_COL_MAJOR;
    *out_layout = lay;
    return sprs_hip_csmat_mulacc_dense_f64(lhs, rhs_dev, rhs_rows, k, rhs_layout, ld_rhs, out_dev, lhs->rows, lay,
                                           lay == SPRS_HIP_ROW_MAJOR ? k : lhs->rows, 0, stream);
}

int32_t sprs_hip_dense_dot_csmat_f64(const double *lhs_dev, uint64_t lhs_rows, uint64_t lhs_cols, int32_t lhs_layout, uint64_t ld_lhs,
                                     const sprs_hip_csmat *rhs, double *out_dev, int32_t *out_layout, void *stream) {
    clear_error();
    if (!rhs || !out_layout) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    if (lhs_layout != SPRS_HIP_ROW_MAJOR && lhs_layout != SPRS_HIP_COL_MAJOR) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "bad layout tag");
    // csmat.rs:2064-2117: (rhs^T * lhs^T)^T with free transposes — rhs.transpose_view(), lhs.t(), res.reversed_axes()
    // the CSR form of rhs^T: the transpose view of a CSC rhs, or of the (cached) CSC copy of a CSR rhs — either way a view
    const sprs_hip_csmat *src = rhs;
    if (rhs->storage == SPRS_HIP_CSR) {
        sprs_hip_csmat *csc = nullptr;
        SPRS_TRY(other_form(rhs, &csc));
        src = csc;
    }
    // the view stays in the rhs handle (beside as_other): its tile / band plans are built once, not per call
    // The owner's lock is held until the product's kernels are launched: a concurrent refresh / numeric SpGEMM on rhs frees the
    // view in invalidate_caches (behind a device synchronise), and must not do so between the look-up and the launches.
    sprs_hip_csmat *owner = const_cast<sprs_hip_csmat *>(rhs);
    std::lock_guard<std::recursive_mutex> lock(owner->mu);
    if (!owner->t_view) SPRS_TRY(sprs_hip_csmat_transpose_view(src, &owner->t_view));
    sprs_hip_csmat *rt = owner->t_view;
    int32_t lay_t = 0;
    // lhs^T: lhs_cols x lhs_rows, the other layout over the same memory
    const int32_t st = sprs_hip_csmat_mul_dense_f64(rt, lhs_dev, lhs_cols, lhs_rows, lhs_layout == SPRS_HIP_ROW_MAJOR ? SPRS_HIP_COL_MAJOR : SPRS_HIP_ROW_MAJOR,
                                                    ld_lhs, out_dev, &lay_t, stream);
    if (st != SPRS_HIP_OK) return st;
    *out_layout = lay_t == SPRS_HIP_ROW_MAJOR ? SPRS_HIP_COL_MAJOR : SPRS_HIP_ROW_MAJOR;     // reversed_axes()
    return SPRS_HIP_OK;
}

int32_t sprs_hip_dist_comm_count(const sprs_hip_dist *d, int32_t *ranks) {
    clear_error();
    if (!d || !ranks) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    return dist_comm_count(d, ranks);
}

int32_t sprs_hip_dist_peer_handle(sprs_hip_dist *d, void *handle_64_bytes) {
    clear_error();
    if (!d || !handle_64_bytes) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    return dist_peer_handle(d, handle_64_bytes);
}

int32_t sprs_hip_dist_peer_connect(sprs_hip_dist *d, const void *handles, int32_t world) {
    clear_error();
    if (!d || !handles) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    return dist_peer_connect(d, handles, world);
}

int32_t sprs_hip_dist_set_route(sprs_hip_dist *d, int32_t route) {
    clear_error();
    if (!d) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL handle");
    return dist_set_route(d, route);
}

int32_t sprs_hip_dist_route(const sprs_hip_dist *d, int32_t *route) {
    clear_error();
    if (!d || !route) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    return dist_route(d, route);
}

int32_t sprs_hip_dist_free(sprs_hip_dist *d) {
    clear_error();
    dist_free(d);
    return SPRS_HIP_OK;
}

int32_t sprs_hip_triplets_to_cs(uint64_t rows, uint64_t cols, uint64_t n, const void *row_inds_dev, const void *col_inds_dev,
                                int32_t in_idx_bytes, const double *data_dev, int32_t storage, int32_t out_idx_bytes,
                                int32_t out_iptr_bytes, sprs_hip_csmat **out) {
    clear_error();
    if (!out) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (storage != SPRS_HIP_CSR && storage != SPRS_HIP_CSC) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "bad storage tag %d", storage);
    if (!widths_ok(out_iptr_bytes, out_idx_bytes) || (in_idx_bytes != 4 && in_idx_bytes != 8))
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "index widths must be 4 or 8 bytes");
    if (n && (!row_inds_dev || !col_inds_dev || !data_dev)) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL triplet array with n > 0");
    const uint64_t inner = storage == SPRS_HIP_CSR ? cols : rows, outer = storage == SPRS_HIP_CSR ? rows : cols;
    if (out_idx_bytes == 4 && inner > 0xFFFFFFFFull) SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Index type not large enough for this matrix");
    if (out_iptr_bytes == 4 && outer + 1 > 0xFFFFFFFFull) SPRS_FAIL(SPRS_HIP_INDEX_OVERFLOW, "Iptr type not large enough for this matrix");
    return triplets_to_cs(rows, cols, n, row_inds_dev, col_inds_dev, in_idx_bytes, data_dev, storage, out_idx_bytes, out_iptr_bytes, out);
}

namespace {
struct OptionDesc {
    const char *name;
    int64_t sprs_hip::Options::*field;
    int64_t lo, hi;
    bool devtools;
};
const OptionDesc kOptions[] = {
#define SPRS_X(name, def, lo, hi, dev) {#name, &sprs_hip::Options::name, (int64_t)(lo), (int64_t)(hi), (dev) != 0},
    SPRS_HIP_OPTIONS(SPRS_X)
#undef SPRS_X
};
const OptionDesc *find_option(const char *name) {
    for (const OptionDesc &d : kOptions)
        if (!strcmp(name, d.name)) return &d;
    return nullptr;
}
}  // namespace

int32_t sprs_hip_set_option(const char *name, int64_t value) {
    clear_error();
    if (!name) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL name");
    const OptionDesc *d = find_option(name);
    if (!d) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "unknown option '%s'", name);
    if (d->devtools && !DEVTOOLS)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "option '%s' is a developer switch (wrong results / profiling printouts): this library was built without SPRS_HIP_DEVTOOLS", name);
    if (value < d->lo || value > d->hi)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "%s must be in %lld .. %lld", name, (long long)d->lo, (long long)d->hi);
    // the few options whose legal values are not a range
    if (!strcmp(name, "spmv_tile") && value != 0 && value != 2048 && value != 4096) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "spmv_tile must be 0 (auto), 2048 or 4096");
    if (!strcmp(name, "spmv_band_tile") && value != 0 && value != 8192 && value != 16384) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "spmv_band_tile must be 0 (default), 8192 or 16384");
    if (!strcmp(name, "spgemm_tokens") && value == 3) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "spgemm_tokens must be 1, 2 or 4");
    if (!strcmp(name, "spmv_band_split") && value == 1) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "spmv_band_split must be 0 (default) or >= 2");
    options().*(d->field) = value;
    if (!strcmp(name, "pool") && !value) (void)pool_trim();
    return SPRS_HIP_OK;
}

int32_t sprs_hip_get_option(const char *name, int64_t *value) {
    clear_error();
    if (!name || !value) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "NULL argument");
    if (!strcmp(name, "pool_cached_bytes")) {
        *value = (int64_t)pool_cached_bytes();
        return SPRS_HIP_OK;
    }
    if (!strcmp(name, "devtools")) {           // 1 when the library was built with SPRS_HIP_DEVTOOLS
        *value = DEVTOOLS ? 1 : 0;
        return SPRS_HIP_OK;
    }
    const OptionDesc *d = find_option(name);
    if (!d) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "unknown option '%s'", name);
    *value = options().*(d->field);
    return SPRS_HIP_OK;
}

}  // extern "C"
