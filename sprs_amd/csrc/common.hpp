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
    int64_t spmv_kernel = 0;   // 0 auto, 1 tiled, 2 wave-per-row
    int64_t spmv_nt = 1;       // non-temporal loads on indices/data streams
    int64_t spmv_tile = 4096;  // nnz per workgroup tile (2048 or 4096)
    int64_t spmv_xload = 0;    // x gather flavour: 0 plain, 1 non-temporal, 2 sc1 (L1 bypass)
    int64_t spmv_xmask = -1;   // TIMING EXPERIMENTS ONLY: gather x[col & mask] (wrong results unless -1)
};
Options &options();

// ---- SpMV plan: nnz-tile -> first row starting in the tile ----------------
struct SpmvPlan {
    uint32_t tile = 0;              // nnz per tile this plan was built for
    uint64_t ntiles = 0;
    uint64_t *tile_row = nullptr;   // device, ntiles + 1 entries
    std::unordered_map<void *, double *> carry;   // per-stream carry scratch (ntiles doubles)
    void release();
};

}  // namespace sprs_hip

// Device twin of CsMatBase (sprs/src/sparse.rs:94-122).
struct sprs_hip_csmat {
    int32_t storage = SPRS_HIP_CSR;
    uint64_t rows = 0, cols = 0, nnz = 0;
    int32_t iptr_bytes = 8, idx_bytes = 8;
    void *indptr = nullptr;    // device, outer+1 entries, zero based
    void *indices = nullptr;   // device, nnz entries
    double *data = nullptr;    // device, nnz entries
    bool owns = false;
    int device = 0;
    std::mutex mu;             // guards plan
    sprs_hip::SpmvPlan plan;

    uint64_t outer() const { return storage == SPRS_HIP_CSR ? rows : cols; }
    uint64_t inner() const { return storage == SPRS_HIP_CSR ? cols : rows; }
};

namespace sprs_hip {

// spmv.hip
int32_t spmv_f64(sprs_hip_csmat *a, const double *x, double *y, bool accumulate, hipStream_t stream);
// spgemm.hip
int32_t spgemm_f64(const sprs_hip_csmat *a, const sprs_hip_csmat *b, sprs_hip_csmat **c);
int32_t to_other_storage(const sprs_hip_csmat *m, sprs_hip_csmat **out);
// abi.hip
int32_t alloc_csmat(sprs_hip_csmat **out, int32_t storage, uint64_t rows, uint64_t cols, uint64_t nnz,
                    int32_t iptr_bytes, int32_t idx_bytes);

}  // namespace sprs_hip
