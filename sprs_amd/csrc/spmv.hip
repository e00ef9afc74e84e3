// CSR x dense-vector SpMV for gfx950 — device twin of
//   prod::mul_acc_mat_vec_csr      sprs/src/sparse/prod.rs:103-127   (accumulate)
//   prod::csr_mulacc_dense_colmaj  sprs/src/sparse/prod.rs:274-298   (1 rhs column; `&A * &x`)
//
// Design (HBM-bound: 16 B of matrix stream per multiply-add, no reuse, no MFMA):
//
//  * The nnz axis, not the row axis, is what gets partitioned: workgroup c owns
//    the nnz TILE [c*T, (c+1)*T).  Every lane streams `indices` and `data` with
//    16-byte loads that are perfectly coalesced whatever the row lengths are
//    (R-MAT rows run from 0 to 2.3e5 entries), and every workgroup moves the
//    same number of bytes, so the chip is load-balanced by construction.
//  * products a_ik * x_k are staged in LDS (T doubles), then reduced per row
//    SEGMENT of the tile: short segments (< 64 entries) by one lane each,
//    long ones by a whole wave with a shuffle tree.  Row boundaries come from
//    `indptr`, read once, coalesced, and kept in LDS as tile-local offsets.
//  * the part of a tile that belongs to a row which started in an earlier tile
//    (its "head") goes to carry[c]; a second, tiny kernel adds the carries of
//    a long row in tile order.  No float atomics: results are run-to-run
//    deterministic.
//  * tiles are dealt to the 8 XCDs in contiguous chunks (blockIdx -> tile remap)
//    so that neighbouring tiles, which gather neighbouring x entries on banded
//    matrices, share one L2.
//  * indices/data are loaded non-temporally: they are used once, x should keep
//    the L2 / Infinity Cache.
//  * products use a separately rounded multiply and add (-ffp-contract=off),
//    as sprs' MulAcc does (sprs/src/mul_acc.rs:28-30).
#include "common.hpp"

namespace sprs_hip {

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BLOCK = 256;       // 4 waves
constexpr int WAVE = 64;
constexpr int NWAVES = BLOCK / WAVE;
constexpr int SEG_CHUNK = 2048;  // row boundaries staged per pass
constexpr uint32_t LONG_SEG = 64;

template <typename V, bool NT>
__device__ __forceinline__ V stream_load(const V *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// gather of one x entry.  XL = 0: plain load (allocates a 128-byte line in the CU's L1),
// 1: non-temporal, 2: sc1 (served by L2, bypasses L1: no line fill for 8 useful bytes).
template <int XL>
__device__ __forceinline__ double gather_x(const double *p) {
    if constexpr (XL == 1) return __builtin_nontemporal_load(p);
    else if constexpr (XL == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// blockIdx -> tile, contiguous chunk of tiles per XCD (block b runs on XCD b % 8).
__device__ __forceinline__ uint64_t tile_of_block(uint64_t bid, uint64_t ntiles) {
    const uint64_t q = ntiles >> 3, rem = ntiles & 7;
    const uint64_t k = bid & 7, j = bid >> 3;
    return k * q + (k < rem ? k : rem) + j;
}

// ---------------------------------------------------------------------------
// plan: tile_row[c] = first row r with indptr[r] >= c*T   (c = 0..ntiles-1),
//       tile_row[ntiles] = rows
// ---------------------------------------------------------------------------
template <typename PTR>
__global__ void build_tile_rows(const PTR *__restrict__ indptr, uint64_t rows, uint64_t ntiles, uint32_t T,
                                uint64_t *__restrict__ tile_row) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ntiles) return;
    if (c == ntiles) {
        tile_row[c] = rows;
        return;
    }
    const uint64_t target = c * (uint64_t)T;
    uint64_t lo = 0, hi = rows;   // lower_bound over indptr[0..rows)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)indptr[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    tile_row[c] = lo;
}

// ---------------------------------------------------------------------------
// main kernel: one workgroup per nnz tile
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR, int T, bool ACC, bool NT, int XL>
__global__ __launch_bounds__(BLOCK) void spmv_tile_kernel(
    const PTR *__restrict__ indptr, const IDX *__restrict__ indices, const double *__restrict__ data,
    const double *__restrict__ x, double *__restrict__ y, const uint64_t *__restrict__ tile_row,
    double *__restrict__ carry, uint64_t nnz, uint64_t ntiles, uint64_t xmask) {
    constexpr int V = 2;                          // elements per lane per pass (16 B of data)
    constexpr int PASSES = T / (BLOCK * V);
    typedef IDX idx2 __attribute__((ext_vector_type(2)));

    __shared__ __attribute__((aligned(16))) double prod[T];
    __shared__ uint32_t segb[SEG_CHUNK + 1];
    __shared__ uint32_t longlist[T / LONG_SEG + 1];
    __shared__ uint32_t nlong;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1);
    const uint32_t wave = tid / WAVE;
    const uint64_t tile = tile_of_block(blockIdx.x, ntiles);
    const uint64_t base = tile * (uint64_t)T;
    const uint32_t cnt = (nnz - base < (uint64_t)T) ? (uint32_t)(nnz - base) : (uint32_t)T;
    const uint64_t lim = base + cnt;
    const uint64_t R0 = tile_row[tile], R1 = tile_row[tile + 1];
    const uint64_t S = R1 - R0 + 1;               // head + rows starting in this tile

    if (tid == 0) nlong = 0;

    // ---- phase 1: stream the tile, gather x, products -> LDS ---------------
    const IDX *ip = indices + base;
    const double *dp = data + base;
    if (cnt == (uint32_t)T) {
        idx2 ix[PASSES];
        dbl2 av[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t i = p * (BLOCK * V) + tid * V;
            ix[p] = stream_load<idx2, NT>((const idx2 *)(ip + i));
            av[p] = stream_load<dbl2, NT>((const dbl2 *)(dp + i));
        }
        double xv[PASSES][V];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            xv[p][0] = gather_x<XL>(x + ((uint64_t)ix[p][0] & xmask));
            xv[p][1] = gather_x<XL>(x + ((uint64_t)ix[p][1] & xmask));
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t i = p * (BLOCK * V) + tid * V;
            dbl2 pr;
            pr[0] = av[p][0] * xv[p][0];
            pr[1] = av[p][1] * xv[p][1];
            *(dbl2 *)&prod[i] = pr;
        }
    } else {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t i = p * (BLOCK * V) + tid * V;
            dbl2 pr = {0.0, 0.0};
            if (i < cnt) pr[0] = dp[i] * x[(uint64_t)ip[i] & xmask];
            if (i + 1 < cnt) pr[1] = dp[i + 1] * x[(uint64_t)ip[i + 1] & xmask];
            *(dbl2 *)&prod[i] = pr;
        }
    }

    // ---- phase 2: per-segment sums -----------------------------------------
    // segment 0 = head (tail of row R0-1), segment j>=1 = row R0+j-1.
    // boundary B[0] = 0, B[j] = min(indptr[R0+j-1], lim) - base   (j = 1..S)
    for (uint64_t j0 = 0; j0 < S; j0 += SEG_CHUNK) {
        const uint32_t n = (S - j0 < (uint64_t)SEG_CHUNK) ? (uint32_t)(S - j0) : (uint32_t)SEG_CHUNK;
        for (uint32_t t = tid; t <= n; t += BLOCK) {
            const uint64_t j = j0 + t;
            uint32_t b = 0;
            if (j != 0) {
                const uint64_t v = (uint64_t)indptr[R0 + j - 1];
                b = (uint32_t)((v < lim ? v : lim) - base);
            }
            segb[t] = b;
        }
        __syncthreads();   // prod[], segb[], nlong visible

        for (uint32_t t = tid; t < n; t += BLOCK) {
            const uint32_t a = segb[t], b = segb[t + 1];
            const uint32_t len = b - a;
            if (len >= LONG_SEG) {
                longlist[atomicAdd(&nlong, 1u)] = t;
            } else {
                double s = 0.0;
                for (uint32_t k = a; k < b; ++k) s += prod[k];
                const uint64_t j = j0 + t;
                if (j == 0) {
                    carry[tile] = s;
                } else {
                    const uint64_t r = R0 + j - 1;
                    if constexpr (ACC) {
                        if (len) y[r] = y[r] + s;   // empty rows keep y[r] untouched (prod.rs:120-126)
                    } else {
                        y[r] = s;
                    }
                }
            }
        }
        __syncthreads();   // longlist complete

        const uint32_t nl = nlong;
        for (uint32_t q = wave; q < nl; q += NWAVES) {
            const uint32_t t = longlist[q];
            const uint32_t a = segb[t], b = segb[t + 1];
            double s = 0.0;
            for (uint32_t k = a + lane; k < b; k += WAVE) s += prod[k];
            s = wave_sum(s);
            if (lane == 0) {
                const uint64_t j = j0 + t;
                if (j == 0) {
                    carry[tile] = s;
                } else {
                    const uint64_t r = R0 + j - 1;
                    if constexpr (ACC) y[r] = y[r] + s;
                    else y[r] = s;
                }
            }
        }
        __syncthreads();   // everyone done with segb/longlist before the next chunk
        if (tid == 0) nlong = 0;
    }
}

// ---------------------------------------------------------------------------
// fix-up: a row that spans several tiles gets the heads of the later tiles
// added in tile order (deterministic).  One thread per tile.
// ---------------------------------------------------------------------------
template <typename PTR>
__global__ void spmv_carry_kernel(const PTR *__restrict__ indptr, const uint64_t *__restrict__ tile_row,
                                  const double *__restrict__ carry, double *__restrict__ y, uint64_t ntiles,
                                  uint32_t T) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c + 1 >= ntiles) return;
    const uint64_t R0 = tile_row[c], R1 = tile_row[c + 1];
    if (R1 == R0) return;                                    // no row starts in tile c
    if ((uint64_t)indptr[R1] <= (c + 1) * (uint64_t)T) return;   // last row ends inside tile c
    double acc = 0.0;
    for (uint64_t d = c + 1; d < ntiles && tile_row[d] == R1; ++d) acc += carry[d];
    y[R1 - 1] += acc;
}

// ---------------------------------------------------------------------------
// reference-shaped kernel for A/B runs: one wave per row
// ---------------------------------------------------------------------------
template <typename IDX, typename PTR, bool ACC>
__global__ __launch_bounds__(BLOCK) void spmv_rowwave_kernel(const PTR *__restrict__ indptr,
                                                             const IDX *__restrict__ indices,
                                                             const double *__restrict__ data,
                                                             const double *__restrict__ x,
                                                             double *__restrict__ y, uint64_t rows) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t wave_global = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
    const uint64_t nwaves = (uint64_t)gridDim.x * NWAVES;
    for (uint64_t r = wave_global; r < rows; r += nwaves) {
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        double acc = 0.0;
        for (uint64_t p = s + lane; p < e; p += WAVE) acc += data[p] * x[indices[p]];
        acc = wave_sum(acc);
        if (lane == 0) {
            if constexpr (ACC) {
                if (e > s) y[r] = y[r] + acc;
            } else {
                y[r] = acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int32_t ensure_plan(sprs_hip_csmat *a, uint32_t T, hipStream_t stream, double **carry_out) {
    std::lock_guard<std::mutex> lock(a->mu);
    SpmvPlan &pl = a->plan;
    if (pl.tile != T) {
        pl.release();
        pl.tile = T;
        pl.ntiles = (a->nnz + T - 1) / T;
        if (pl.ntiles) {
            SPRS_TRY_HIP(hipMalloc((void **)&pl.tile_row, (pl.ntiles + 1) * sizeof(uint64_t)));
            const uint64_t n = pl.ntiles + 1;
            const dim3 grid((unsigned)((n + 255) / 256)), block(256);
            if (a->iptr_bytes == 8)
                hipLaunchKernelGGL(build_tile_rows<uint64_t>, grid, block, 0, stream, (const uint64_t *)a->indptr,
                                   a->rows, pl.ntiles, T, pl.tile_row);
            else
                hipLaunchKernelGGL(build_tile_rows<uint32_t>, grid, block, 0, stream, (const uint32_t *)a->indptr,
                                   a->rows, pl.ntiles, T, pl.tile_row);
            SPRS_TRY_HIP(hipGetLastError());
            // other streams may use the plan next: make it globally visible once
            SPRS_TRY_HIP(hipStreamSynchronize(stream));
        }
    }
    *carry_out = nullptr;
    if (pl.ntiles) {
        auto it = pl.carry.find((void *)stream);
        if (it == pl.carry.end()) {
            double *c = nullptr;
            SPRS_TRY_HIP(hipMalloc((void **)&c, pl.ntiles * sizeof(double)));
            it = pl.carry.emplace((void *)stream, c).first;
        }
        *carry_out = it->second;
    }
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR, int T>
static int32_t launch_tiled(sprs_hip_csmat *a, const double *x, double *y, bool acc, bool nt, double *carry,
                            hipStream_t stream) {
    const SpmvPlan &pl = a->plan;
    if (pl.ntiles > 0x7fffffffull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "too many tiles for one launch");
    const dim3 grid((unsigned)pl.ntiles), block(BLOCK);
    const IDX *ix = (const IDX *)a->indices;
    const PTR *ip = (const PTR *)a->indptr;
#define SPRS_LAUNCH(ACC_, NT_, XL_)                                                                              \
    hipLaunchKernelGGL((spmv_tile_kernel<IDX, PTR, T, ACC_, NT_, XL_>), grid, block, 0, stream, ip, ix, a->data, x, y, \
                       pl.tile_row, carry, a->nnz, pl.ntiles, (uint64_t)options().spmv_xmask)
#define SPRS_LAUNCH_XL(ACC_, NT_)                 \
    do {                                          \
        if (xl == 2) SPRS_LAUNCH(ACC_, NT_, 2);   \
        else if (xl == 1) SPRS_LAUNCH(ACC_, NT_, 1); \
        else SPRS_LAUNCH(ACC_, NT_, 0);           \
    } while (0)
    const int xl = (int)options().spmv_xload;
    if (acc) {
        if (nt) SPRS_LAUNCH_XL(true, true);
        else SPRS_LAUNCH_XL(true, false);
    } else {
        if (nt) SPRS_LAUNCH_XL(false, true);
        else SPRS_LAUNCH_XL(false, false);
    }
#undef SPRS_LAUNCH_XL
#undef SPRS_LAUNCH
    SPRS_TRY_HIP(hipGetLastError());
    if (pl.ntiles > 1) {
        const dim3 g2((unsigned)((pl.ntiles + 255) / 256)), b2(256);
        hipLaunchKernelGGL(spmv_carry_kernel<PTR>, g2, b2, 0, stream, ip, pl.tile_row, carry, y, pl.ntiles,
                           (uint32_t)T);
        SPRS_TRY_HIP(hipGetLastError());
    }
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR>
static int32_t launch_rowwave(sprs_hip_csmat *a, const double *x, double *y, bool acc, hipStream_t stream) {
    uint64_t blocks = (a->rows + NWAVES - 1) / NWAVES;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid((unsigned)blocks), block(BLOCK);
    if (acc)
        hipLaunchKernelGGL((spmv_rowwave_kernel<IDX, PTR, true>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, x, y, a->rows);
    else
        hipLaunchKernelGGL((spmv_rowwave_kernel<IDX, PTR, false>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, x, y, a->rows);
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR>
static int32_t dispatch(sprs_hip_csmat *a, const double *x, double *y, bool acc, hipStream_t stream) {
    const Options &o = options();
    if (o.spmv_kernel == 2) return launch_rowwave<IDX, PTR>(a, x, y, acc, stream);
    const uint32_t T = (uint32_t)o.spmv_tile;
    double *carry = nullptr;
    SPRS_TRY(ensure_plan(a, T, stream, &carry));
    const bool nt = o.spmv_nt != 0;
    if (T == 2048) return launch_tiled<IDX, PTR, 2048>(a, x, y, acc, nt, carry, stream);
    return launch_tiled<IDX, PTR, 4096>(a, x, y, acc, nt, carry, stream);
}

int32_t spmv_f64(sprs_hip_csmat *a, const double *x, double *y, bool accumulate, hipStream_t stream) {
    if (a->rows == 0) return SPRS_HIP_OK;
    if (a->nnz == 0) {
        // all rows empty: accumulate leaves y alone, the operator form yields zeros (csmat.rs:2137)
        if (!accumulate) SPRS_TRY_HIP(hipMemsetAsync(y, 0, a->rows * sizeof(double), stream));
        return SPRS_HIP_OK;
    }
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return dispatch<uint64_t, uint64_t>(a, x, y, accumulate, stream);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return dispatch<uint32_t, uint64_t>(a, x, y, accumulate, stream);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return dispatch<uint64_t, uint32_t>(a, x, y, accumulate, stream);
    return dispatch<uint32_t, uint32_t>(a, x, y, accumulate, stream);
}

}  // namespace sprs_hip
