// CSR x dense-vector SpMV for gfx950 — device twin of
//   prod::mul_acc_mat_vec_csr      sprs/src/sparse/prod.rs:103-127   (accumulate)
//   prod::csr_mulacc_dense_colmaj  sprs/src/sparse/prod.rs:274-298   (1 rhs column; `&A * &x`)
//
// Design (HBM-bound: 16 B of matrix stream per multiply-add, no reuse, no MFMA):
//
//  1. THE TILE KERNEL.  The nnz axis, not the row axis, is partitioned: workgroup c
//     owns the nnz TILE [c*T, (c+1)*T).  Every lane streams `indices` and `data` with
//     16-byte non-temporal loads that are perfectly coalesced whatever the row lengths
//     are (R-MAT rows run from 0 to 2.3e5 entries), and every workgroup moves the same
//     number of bytes: load balance by construction.  Products a_ik * x_k are staged
//     in LDS (T doubles) and reduced per row SEGMENT of the tile — short segments
//     (< 64 entries) by one lane each, long ones by a whole wave with a shuffle tree;
//     row boundaries come from `indptr`, read once, coalesced, kept in LDS as
//     tile-local offsets.  The part of a tile that belongs to a row which started in an
//     earlier tile goes to carry[c]; a tiny second kernel adds the carries of a long
//     row in tile order.  No float atomics anywhere: run-to-run deterministic.
//
//  2. THE XCD-SLICED PLAN.  On power-law matrices the gathers x[col], not the stream,
//     set the time: rocprofv3 counters on R-MAT 10M show 286 M L1->L2 gather requests
//     per SpMV of which 42 % miss the 4 MiB L2, i.e. ~21 GB of fabric traffic for
//     5.4 GB of algorithmic bytes (profiles/).  MI355X has EIGHT private L2s (one per
//     XCD); with a plain launch all eight cache the same hot x lines.  The plan
//     therefore splits the matrix once, at first use: rows shorter than `split` stay
//     in a CSR piece of their own; for the long rows (90 % of the entries) slice s
//     holds the entries whose x line (col >> 4) hashes to s, and slice s is
//     processed by workgroups with blockIdx % 8 == s, i.e. on XCD s.  Each L2 then
//     serves 1/8 of x: eight times the effective cache.  Slice results are per-row
//     partials, summed in slice order by a last small kernel (deterministic).
//     The split is a pure re-layout in HBM (the handle keeps its CSR arrays);
//     correctness never depends on where a workgroup runs.  The copies carry column ids
//     RELABELLED by popularity class, and x is gathered into that order at the start of every
//     SpMV, so that the 16 x entries of an L2 line are equally popular (see the rl_* kernels).
//
//  3. Products use a separately rounded multiply and add (-ffp-contract=off), as sprs'
//     MulAcc does (sprs/src/mul_acc.rs:28-30); only the summation order inside a row
//     differs from the reference (tree instead of left-to-right).
#include "spmv_shared.hpp"

namespace sprs_hip {

struct TileArgs {
    const void *indptr;          // PTR[rows + 1]
    const void *indices;         // IDX[nnz]
    const double *data;
    const uint16_t *pos;         // nnz, or NULL: tile-local position of an entry whose tile was sorted by column
    const uint64_t *tile_row;    // ntiles + 1
    double *carry;               // ntiles
    double *y;
    uint64_t nnz, ntiles;
};

struct SlicedArgs {
    TileArgs p[XCS_SLICES];
};


// blockIdx -> tile, contiguous chunk of tiles per XCD (block b runs on XCD b % 8):
// neighbouring tiles gather neighbouring x entries on banded matrices and share one L2.
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
__global__ void build_tile_rows(const PTR *__restrict__ indptr, uint64_t rows, uint64_t ntiles, uint32_t TILE,
                                uint64_t *__restrict__ tile_row) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ntiles) return;
    if (c == ntiles) {
        tile_row[c] = rows;
        return;
    }
    const uint64_t target = c * (uint64_t)TILE;
    uint64_t lo = 0, hi = rows;   // lower_bound over indptr[0..rows)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)indptr[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    tile_row[c] = lo;
}

// ---------------------------------------------------------------------------
// one workgroup, one nnz tile
// ---------------------------------------------------------------------------
// developer builds only (option spmv_xmask): gathers forced into a narrow window of x, a TIMING experiment with wrong
// results; the release library compiles it out
__device__ __forceinline__ uint64_t xm(uint64_t col, uint64_t xmask) {
    if constexpr (DEVTOOLS) return col & xmask;
    else return col;
}

template <typename IDX, typename PTR, bool ACC, int TILE>
__device__ __forceinline__ void tile_body(const TileArgs &a, const double *__restrict__ x, uint64_t tile,
                                          uint64_t xmask) {
    constexpr int V = 2;                          // elements per lane per pass (16 B of data)
    constexpr int PASSES = TILE / (BLOCK * V);
    typedef IDX idx2 __attribute__((ext_vector_type(2)));

    __shared__ __attribute__((aligned(16))) double prod[TILE];
    __shared__ uint32_t segb[SEG_CHUNK + 1];
    __shared__ uint32_t longlist[TILE / LONG_SEG + 1];
    __shared__ uint32_t nlong;

    const PTR *__restrict__ indptr = (const PTR *)a.indptr;
    double *__restrict__ y = a.y;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1);
    const uint32_t wave = tid / WAVE;
    const uint64_t base = tile * (uint64_t)TILE;
    const uint32_t cnt = (a.nnz - base < (uint64_t)TILE) ? (uint32_t)(a.nnz - base) : (uint32_t)TILE;
    const uint64_t lim = base + cnt;
    const uint64_t R0 = a.tile_row[tile], R1 = a.tile_row[tile + 1];
    const uint64_t S = R1 - R0 + 1;               // head + rows starting in this tile

    if (tid == 0) nlong = 0;

    // ---- phase 1: stream the tile, gather x, products -> LDS ---------------
    const IDX *ip = (const IDX *)a.indices + base;
    const double *dp = a.data + base;
    if (cnt == (uint32_t)TILE) {
        idx2 ix[PASSES];
        dbl2 av[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t i = p * (BLOCK * V) + tid * V;
            ix[p] = __builtin_nontemporal_load((const idx2 *)(ip + i));   // used once: keep L2 for x
            av[p] = __builtin_nontemporal_load((const dbl2 *)(dp + i));
        }
        double xv[PASSES][V];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            xv[p][0] = x[xm((uint64_t)ix[p][0], xmask)];
            xv[p][1] = x[xm((uint64_t)ix[p][1], xmask)];
        }
        if (a.pos) {
            // the tile's entries are stored sorted by column (plan copies only): lanes next to each
            // other gather from the same x lines; products return to their row-major slot in LDS
            const uint16_t *pp = a.pos + base;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const uint32_t i = p * (BLOCK * V) + tid * V;
                const u16x2 q = __builtin_nontemporal_load((const u16x2 *)(pp + i));
                prod[q[0]] = av[p][0] * xv[p][0];
                prod[q[1]] = av[p][1] * xv[p][1];
            }
        } else {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const uint32_t i = p * (BLOCK * V) + tid * V;
                dbl2 pr;
                pr[0] = av[p][0] * xv[p][0];
                pr[1] = av[p][1] * xv[p][1];
                *(dbl2 *)&prod[i] = pr;
            }
        }
    } else {
        const uint16_t *pp = a.pos ? a.pos + base : nullptr;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t i = p * (BLOCK * V) + tid * V;
            const double p0 = i < cnt ? dp[i] * x[xm((uint64_t)ip[i], xmask)] : 0.0;
            const double p1 = i + 1 < cnt ? dp[i + 1] * x[xm((uint64_t)ip[i + 1], xmask)] : 0.0;
            if (pp) {
                if (i < cnt) prod[pp[i]] = p0;
                if (i + 1 < cnt) prod[pp[i + 1]] = p1;
            } else {
                prod[i] = p0;
                prod[i + 1] = p1;
            }
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
            const uint32_t sa = segb[t], sb = segb[t + 1];
            const uint32_t len = sb - sa;
            if (len >= LONG_SEG) {
                longlist[atomicAdd(&nlong, 1u)] = t;
            } else {
                double s = 0.0;
                for (uint32_t k = sa; k < sb; ++k) s += prod[k];
                const uint64_t j = j0 + t;
                if (j == 0) {
                    a.carry[tile] = s;
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
            const uint32_t sa = segb[t], sb = segb[t + 1];
            double s = 0.0;
            for (uint32_t k = sa + lane; k < sb; k += WAVE) s += prod[k];
            s = wave_sum(s);
            if (lane == 0) {
                const uint64_t j = j0 + t;
                if (j == 0) {
                    a.carry[tile] = s;
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

template <typename IDX, typename PTR, bool ACC, int TILE>
__global__ __launch_bounds__(BLOCK) void spmv_tile_kernel(TileArgs a, const double *__restrict__ x, uint64_t xmask) {
    tile_body<IDX, PTR, ACC, TILE>(a, x, tile_of_block(blockIdx.x, a.ntiles), xmask);
}

// Sliced launch: workgroup b works on slice b % 8 — the dispatcher places block b on
// XCD b % 8, so slice s's x lines live in ONE L2.  (Placement only affects speed.)
template <typename IDX, int TILE>
__global__ __launch_bounds__(BLOCK) void spmv_sliced_kernel(const SlicedArgs *__restrict__ sa,
                                                            const double *__restrict__ x, uint64_t xmask) {
    const uint32_t s = blockIdx.x & (XCS_SLICES - 1);
    const uint64_t tile = blockIdx.x >> 3;
    const TileArgs a = sa->p[s];
    if (tile >= a.ntiles) return;
    tile_body<IDX, uint64_t, false, TILE>(a, x, tile, xmask);
}

// ---------------------------------------------------------------------------
// fix-up: a row that spans several tiles gets the heads of the later tiles
// added in tile order (deterministic).  One thread per tile.
// ---------------------------------------------------------------------------
template <typename PTR>
__device__ __forceinline__ void carry_body(const TileArgs &a, uint64_t c, uint32_t TILE) {
    if (c + 1 >= a.ntiles) return;
    const uint64_t R0 = a.tile_row[c], R1 = a.tile_row[c + 1];
    if (R1 == R0) return;                                                        // no row starts in tile c
    if ((uint64_t)((const PTR *)a.indptr)[R1] <= (c + 1) * (uint64_t)TILE) return;   // last row ends inside tile c
    double acc = 0.0;
    for (uint64_t d = c + 1; d < a.ntiles && a.tile_row[d] == R1; ++d) acc += a.carry[d];
    a.y[R1 - 1] += acc;
}

template <typename PTR>
__global__ void spmv_carry_kernel(TileArgs a, uint32_t TILE) {
    carry_body<PTR>(a, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, TILE);
}

__global__ void spmv_sliced_carry_kernel(const SlicedArgs *__restrict__ sa, uint32_t TILE) {
    const TileArgs a = sa->p[blockIdx.y];
    carry_body<uint64_t>(a, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, TILE);
}

// y[long_rows[j]] (+)= sum over slices, in slice order
template <bool ACC>
__global__ void xcs_reduce_kernel(const double *__restrict__ partial, const uint64_t *__restrict__ long_rows,
                                  double *__restrict__ y, uint64_t n_long) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_long) return;
    double s = partial[j];
#pragma unroll
    for (int k = 1; k < XCS_SLICES; ++k) s += partial[(uint64_t)k * n_long + j];
    const uint64_t r = long_rows[j];
    if constexpr (ACC) y[r] = y[r] + s;
    else y[r] = s;
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
// building the XCD-sliced plan (one-time re-layout, all on the device)
// ---------------------------------------------------------------------------
template <typename PTR>
__global__ void xcs_classify_kernel(const PTR *__restrict__ indptr, uint64_t rows, uint64_t split,
                                    uint64_t *__restrict__ short_len, uint64_t *__restrict__ long_flag) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t len = (uint64_t)indptr[r + 1] - (uint64_t)indptr[r];
    const bool is_long = len >= split;
    short_len[r] = is_long ? 0 : len;
    long_flag[r] = is_long ? 1 : 0;
}

// short rows: copied into their own CSR piece (long rows become empty rows of it)
template <typename IDX, typename PTR, typename CIDX>
__global__ void xcs_fill_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                const double *__restrict__ data, uint64_t rows, const uint64_t *__restrict__ long_flag,
                                const uint64_t *__restrict__ short_ptr, const uint64_t *__restrict__ long_pos,
                                PTR *__restrict__ s_indptr, CIDX *__restrict__ s_indices, double *__restrict__ s_data,
                                uint64_t *__restrict__ long_rows, const uint32_t *__restrict__ perm) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    s_indptr[r] = (PTR)short_ptr[r];
    if (r == rows) return;
    if (long_flag[r]) {
        long_rows[long_pos[r]] = r;
        return;
    }
    const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
    uint64_t d = short_ptr[r];
    for (uint64_t p = s; p < e; ++p, ++d) {
        s_indices[d] = perm ? (CIDX)perm[indices[p]] : (CIDX)indices[p];
        s_data[d] = data[p];
    }
}

// long rows: entries per (row, slice); cnt is slice-major: cnt[s * n_long + j]
template <typename IDX, typename PTR>
__global__ __launch_bounds__(BLOCK) void xcs_count_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                          const uint64_t *__restrict__ long_rows, uint64_t n_long,
                                                          uint64_t *__restrict__ cnt, const uint32_t *__restrict__ perm) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t w0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * NWAVES;
    for (uint64_t j = w0; j < n_long; j += nw) {
        const uint64_t r = long_rows[j];
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        uint32_t c[XCS_SLICES] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint64_t p0 = s; p0 < e; p0 += WAVE) {
            const uint64_t p = p0 + lane;
            const uint32_t sl = p < e ? x_slice(perm ? (uint64_t)perm[indices[p]] : (uint64_t)indices[p]) : XCS_SLICES;
#pragma unroll
            for (int k = 0; k < XCS_SLICES; ++k) c[k] += (uint32_t)__popcll(__ballot(sl == (uint32_t)k));
        }
        if (lane < XCS_SLICES) {
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < XCS_SLICES; ++k) mine = (lane == (uint32_t)k) ? c[k] : mine;
            cnt[(uint64_t)lane * n_long + j] = mine;
        }
    }
}

struct SliceOut {
    const uint64_t *ptr[XCS_SLICES];   // per-slice indptr (n_long + 1)
    void *indices[XCS_SLICES];
    double *data[XCS_SLICES];
};

// stable partition of every long row into its 8 slices (column order kept)
template <typename IDX, typename PTR, typename CIDX>
__global__ __launch_bounds__(BLOCK) void xcs_scatter_kernel(const PTR *__restrict__ indptr,
                                                            const IDX *__restrict__ indices,
                                                            const double *__restrict__ data,
                                                            const uint64_t *__restrict__ long_rows, uint64_t n_long,
                                                            SliceOut out, const uint32_t *__restrict__ perm) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint64_t w0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
    const uint64_t nw = (uint64_t)gridDim.x * NWAVES;
    for (uint64_t j = w0; j < n_long; j += nw) {
        const uint64_t r = long_rows[j];
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        uint64_t base[XCS_SLICES];
#pragma unroll
        for (int k = 0; k < XCS_SLICES; ++k) base[k] = out.ptr[k][j];
        for (uint64_t p0 = s; p0 < e; p0 += WAVE) {
            const uint64_t p = p0 + lane;
            const bool valid = p < e;
            const uint64_t c = valid ? (perm ? (uint64_t)perm[indices[p]] : (uint64_t)indices[p]) : 0ull;
            const double v = valid ? data[p] : 0.0;
            const uint32_t sl = valid ? x_slice(c) : XCS_SLICES;
#pragma unroll
            for (int k = 0; k < XCS_SLICES; ++k) {
                const unsigned long long m = __ballot(sl == (uint32_t)k);
                if (sl == (uint32_t)k) {
                    const uint64_t pos = base[k] + (uint64_t)__popcll(m & below);
                    ((CIDX *)out.indices[k])[pos] = (CIDX)c;
                    out.data[k][pos] = v;
                }
                base[k] += (uint64_t)__popcll(m);
            }
        }
    }
}

// Sort the entries of every tile of a plan-owned piece by column (ties cannot occur inside a
// row; across rows the original position breaks them), remembering where each entry came from.
// One workgroup per tile, bitonic sort of 64-bit keys (col << 16 | position) in LDS.
// Why: the SpMV is bound by the number of L2 line accesses its gathers make (one 128-byte line
// per 8 useful bytes, ~0.9 per entry).  In a column-sorted tile the lanes of a wave read
// neighbouring — often identical — lines, which the texture addresser merges into one access.
template <typename CIDX, int TILE>
__global__ __launch_bounds__(BLOCK) void sort_tiles_kernel(CIDX *__restrict__ indices, double *__restrict__ data,
                                                           uint16_t *__restrict__ pos, uint64_t nnz) {
    __shared__ unsigned long long key[TILE];
    __shared__ double val[TILE];
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint32_t cnt = (nnz - base < (uint64_t)TILE) ? (uint32_t)(nnz - base) : (uint32_t)TILE;
    for (uint32_t i = tid; i < (uint32_t)TILE; i += BLOCK) {
        key[i] = i < cnt ? (((unsigned long long)indices[base + i] << 16) | i) : ~0ull;
        val[i] = i < cnt ? data[base + i] : 0.0;
    }
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= (uint32_t)TILE; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < (uint32_t)TILE; i += BLOCK) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long ki = key[i], kl = key[l];
                    const bool asc = (i & k2) == 0;
                    if ((ki > kl) == asc) {
                        key[i] = kl;
                        key[l] = ki;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < cnt; i += BLOCK) {
        const unsigned long long k = key[i];
        const uint32_t from = (uint32_t)(k & 0xFFFFull);
        indices[base + i] = (CIDX)(k >> 16);
        data[base + i] = val[from];
        pos[base + i] = (uint16_t)from;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename PTR>
static int32_t make_tile_rows(CsrPiece &pc, uint32_t TILE, hipStream_t stream) {
    pc.ntiles = (pc.nnz + TILE - 1) / TILE;
    if (!pc.ntiles) return SPRS_HIP_OK;
    if (pc.ntiles * XCS_SLICES > 0x7fffffffull) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "too many tiles for one launch");
    SPRS_TRY_HIP(hipMalloc((void **)&pc.tile_row, (pc.ntiles + 1) * sizeof(uint64_t)));
    const uint64_t n = pc.ntiles + 1;
    hipLaunchKernelGGL(build_tile_rows<PTR>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const PTR *)pc.indptr, pc.rows, pc.ntiles, TILE, pc.tile_row);
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

// re-layout of the long rows / short rows into plan-owned pieces whose column ids are CIDX
template <typename IDX, typename PTR, typename CIDX>
static int32_t build_sliced(sprs_hip_csmat *a, uint64_t nnz_short, uint64_t n_long, TmpBuf &long_flag,
                            TmpBuf &short_ptr, TmpBuf &long_pos, hipStream_t stream) {
    SpmvPlan &pl = a->plan;
    const uint64_t rows = a->rows;
    const PTR *ip = (const PTR *)a->indptr;
    const IDX *ix = (const IDX *)a->indices;
    // ---- column relabelling (see rl_* kernels) ---------------------------------
    const Options &o = options();
    pl.cols = a->cols;
    if (o.spmv_relabel != 2 && a->cols <= 0xFFFFFFFFull && a->nnz) {
        SPRS_TRY(build_column_labels<IDX>(ix, a->nnz, a->cols, stream, &pl.perm));
    }
    // ---- short part + list of long rows --------------------------------------
    pl.xcs = true;
    pl.n_long = n_long;
    pl.idx_bytes = (int)sizeof(CIDX);
    pl.main.rows = rows;
    pl.main.nnz = nnz_short;
    pl.main.owns = true;
    SPRS_TRY_HIP(hipMalloc(&pl.main.indptr, (rows + 1) * sizeof(PTR)));
    SPRS_TRY_HIP(hipMalloc(&pl.main.indices, (nnz_short ? nnz_short : 4) * sizeof(CIDX)));
    SPRS_TRY_HIP(hipMalloc((void **)&pl.main.data, (nnz_short ? nnz_short : 2) * sizeof(double)));
    SPRS_TRY_HIP(hipMalloc((void **)&pl.long_rows, n_long * sizeof(uint64_t)));
    hipLaunchKernelGGL((xcs_fill_kernel<IDX, PTR, CIDX>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream,
                       ip, ix, a->data, rows, long_flag.u64(), short_ptr.u64(), long_pos.u64(), (PTR *)pl.main.indptr,
                       (CIDX *)pl.main.indices, pl.main.data, pl.long_rows, pl.perm);
    SPRS_TRY_HIP(hipGetLastError());

    // ---- long part: count, scan, scatter ------------------------------------------
    TmpBuf cnt;
    SPRS_TRY_HIP(cnt.alloc(XCS_SLICES * n_long * 8));
    uint64_t wblocks = (n_long + NWAVES - 1) / NWAVES;
    if (wblocks > 256 * 64) wblocks = 256 * 64;
    hipLaunchKernelGGL((xcs_count_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(BLOCK), 0, stream, ip, ix,
                       pl.long_rows, n_long, cnt.u64(), pl.perm);
    SPRS_TRY_HIP(hipGetLastError());
    SliceOut so;
    for (int s = 0; s < XCS_SLICES; ++s) {
        CsrPiece &sl = pl.slice[s];
        sl.rows = n_long;
        sl.owns = true;
        SPRS_TRY_HIP(hipMalloc(&sl.indptr, (n_long + 1) * sizeof(uint64_t)));
        SPRS_TRY(exclusive_scan_u64(cnt.u64() + (uint64_t)s * n_long, (uint64_t *)sl.indptr, n_long, stream));
        SPRS_TRY_HIP(hipMemcpy(&sl.nnz, (uint64_t *)sl.indptr + n_long, 8, hipMemcpyDeviceToHost));
        SPRS_TRY_HIP(hipMalloc(&sl.indices, (sl.nnz ? sl.nnz : 4) * sizeof(CIDX)));
        SPRS_TRY_HIP(hipMalloc((void **)&sl.data, (sl.nnz ? sl.nnz : 2) * sizeof(double)));
        so.ptr[s] = (const uint64_t *)sl.indptr;
        so.indices[s] = sl.indices;
        so.data[s] = sl.data;
    }
    hipLaunchKernelGGL((xcs_scatter_kernel<IDX, PTR, CIDX>), dim3((unsigned)wblocks), dim3(BLOCK), 0, stream, ip, ix,
                       a->data, pl.long_rows, n_long, so, pl.perm);
    SPRS_TRY_HIP(hipGetLastError());

    SPRS_TRY(make_tile_rows<PTR>(pl.main, pl.tile, stream));
    pl.slice_tile_off[0] = 0;
    for (int s = 0; s < XCS_SLICES; ++s) {
        SPRS_TRY(make_tile_rows<uint64_t>(pl.slice[s], pl.tile, stream));
        pl.slice_tile_off[s + 1] = pl.slice_tile_off[s] + pl.slice[s].ntiles;
    }
    if (options().spmv_sort_tiles) {
        auto sort_piece = [&](CsrPiece &pc) -> int32_t {
            if (!pc.ntiles) return SPRS_HIP_OK;
            SPRS_TRY_HIP(hipMalloc((void **)&pc.pos, pc.nnz * sizeof(uint16_t) + 16));
            const dim3 grid((unsigned)pc.ntiles), block(BLOCK);
            if (pl.tile == 2048)
                hipLaunchKernelGGL((sort_tiles_kernel<CIDX, 2048>), grid, block, 0, stream, (CIDX *)pc.indices, pc.data,
                                   pc.pos, pc.nnz);
            else
                hipLaunchKernelGGL((sort_tiles_kernel<CIDX, 4096>), grid, block, 0, stream, (CIDX *)pc.indices, pc.data,
                                   pc.pos, pc.nnz);
            SPRS_TRY_HIP(hipGetLastError());
            return SPRS_HIP_OK;
        };
        SPRS_TRY(sort_piece(pl.main));
        for (int s = 0; s < XCS_SLICES; ++s) SPRS_TRY(sort_piece(pl.slice[s]));
    }
    return SPRS_HIP_OK;
}

// the options a plan depends on: a handle rebuilds its plan when one of them changes
static uint64_t plan_signature(const Options &o) {
    const int64_t v[] = {o.spmv_xcs, o.spmv_xcs_split, o.spmv_xcs_idx32, o.spmv_sort_tiles, o.spmv_relabel, o.spmv_tile, o.spmv_band,
                         o.spmv_band_hot, o.spmv_band_tile, o.spmv_band_phases, o.spmv_band_split, o.spmv_band_rounds, o.spmv_band_cold_tiles, o.spmv_band_hot_run, o.spmv_band_share, o.spmv_band_balance};
    uint64_t h = 0xcbf29ce484222325ull;
    for (int64_t x : v) h = (h ^ (uint64_t)x) * 0x100000001b3ull;
    return h | 1ull;
}

// light: the handle's first multiply — a plan that COPIES the matrix (banded, XCD-sliced) is not built yet unless the options
// force it; pl.light then says that the next multiply should come back here.  A handle that multiplies once (`&a * &x` on a
// temporary) never pays the ~0.1 s of a copy plan; one that iterates pays it at its second multiply.
template <typename IDX, typename PTR>
static int32_t build_plan(sprs_hip_csmat *a, hipStream_t stream, bool light = false) {
    SpmvPlan &pl = a->plan;
    const Options &o = options();
    pl.release();
    pl.opt_sig = plan_signature(o);
    pl.idx_bytes = (int)sizeof(IDX);
    pl.light = false;
    const uint64_t rows = a->rows, nnz = a->nnz;
    const PTR *ip = (const PTR *)a->indptr;

    // auto: worth it only when x is much larger than one L2 and the matrix is big enough to amortise
    bool want = !a->one_shot && (o.spmv_xcs == 1 || (o.spmv_xcs == 0 && a->cols * 8 >= (16ull << 20) && nnz >= (1ull << 22)));
    // the banded plan (spmv_band.hip) takes such matrices when it applies (its own test of the row lengths)
    // (it pays from smaller x on than the XCD-sliced plan: R-MAT 1M, x = 8 MB, cold caches: 0.101 vs 0.156 ms, profiles/r02p)
    bool want_band = !a->one_shot && (o.spmv_band == 1 || (o.spmv_band == 0 && o.spmv_xcs != 2 && a->cols * 8 >= (4ull << 20) && nnz >= (1ull << 22)));
    if (light) {
        const bool auto_band = want_band && o.spmv_band != 1, auto_xcs = want && o.spmv_xcs != 1;
        pl.light = auto_band || auto_xcs;
        if (auto_band) want_band = false;
        if (auto_xcs) want = false;
    }
    if (want_band) {
        const int32_t st = band_build(a, stream, &pl.band);
        // auto mode: a matrix the banded plan cannot be built for (its temporaries did not fit) keeps the plans below;
        // an explicit spmv_band = 1 reports the failure
        if (st == SPRS_HIP_OUT_OF_MEMORY && o.spmv_band == 0) {
            (void)hipGetLastError();
            clear_error();
            pl.band = nullptr;
        } else if (st != SPRS_HIP_OK) {
            return st;
        }
        if (pl.band) {
            pl.built = true;
            return SPRS_HIP_OK;
        }
    }
    TmpBuf short_len, long_flag, short_ptr, long_pos;
    uint64_t nnz_short = 0, n_long = 0;
    if (want) {
        SPRS_TRY_HIP(short_len.alloc(rows * 8));
        SPRS_TRY_HIP(long_flag.alloc(rows * 8));
        SPRS_TRY_HIP(short_ptr.alloc((rows + 1) * 8));
        SPRS_TRY_HIP(long_pos.alloc((rows + 1) * 8));
        hipLaunchKernelGGL(xcs_classify_kernel<PTR>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, ip,
                           rows, (uint64_t)o.spmv_xcs_split, short_len.u64(), long_flag.u64());
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY(exclusive_scan_u64(short_len.u64(), short_ptr.u64(), rows, stream));
        SPRS_TRY(exclusive_scan_u64(long_flag.u64(), long_pos.u64(), rows, stream));
        SPRS_TRY_HIP(hipMemcpy(&nnz_short, short_ptr.u64() + rows, 8, hipMemcpyDeviceToHost));
        SPRS_TRY_HIP(hipMemcpy(&n_long, long_pos.u64() + rows, 8, hipMemcpyDeviceToHost));
        // auto mode: slice only if the long rows carry most of the entries
        if (n_long == 0 || (o.spmv_xcs == 0 && (nnz - nnz_short) * 2 < nnz)) want = false;
    }
    // tile size: measured on MI355X (profiles/r01f_ab_log.txt) — 4096 for the sliced plan and for
    // plain plans with long rows, 2048 (more workgroups in flight) when rows are short
    if (o.spmv_tile) pl.tile = (uint32_t)o.spmv_tile;
    else pl.tile = (want || nnz > 16 * rows) ? 4096u : 2048u;

    if (!want) {
        pl.main.indptr = a->indptr;
        pl.main.indices = a->indices;
        pl.main.data = a->data;
        pl.main.rows = rows;
        pl.main.nnz = nnz;
        pl.main.owns = false;
        SPRS_TRY(make_tile_rows<PTR>(pl.main, pl.tile, stream));
    } else if (sizeof(IDX) == 8 && o.spmv_xcs_idx32 && a->cols <= 0xFFFFFFFFull) {
        // the plan's own copies hold 32-bit column ids: 12 instead of 16 bytes of stream per entry
        SPRS_TRY((build_sliced<IDX, PTR, uint32_t>(a, nnz_short, n_long, long_flag, short_ptr, long_pos, stream)));
    } else {
        SPRS_TRY((build_sliced<IDX, PTR, IDX>(a, nnz_short, n_long, long_flag, short_ptr, long_pos, stream)));
    }
    SPRS_TRY_HIP(hipStreamSynchronize(stream));   // plan complete and visible to every stream
    pl.built = true;
    return SPRS_HIP_OK;
}

static int32_t get_scratch(SpmvPlan &pl, hipStream_t stream, SpmvScratch **out) {
    auto it = pl.scratch.find((void *)stream);
    if (it == pl.scratch.end()) {
        SpmvScratch sc;
        if (pl.main.ntiles) SPRS_TRY_HIP(hipMalloc((void **)&sc.carry_main, pl.main.ntiles * sizeof(double)));
        if (pl.perm) SPRS_TRY_HIP(hipMalloc((void **)&sc.xp, (pl.cols ? pl.cols : 1) * sizeof(double)));
        if (pl.xcs) {
            const uint64_t nt = pl.slice_tile_off[XCS_SLICES];
            SPRS_TRY_HIP(hipMalloc((void **)&sc.carry_slices, (nt ? nt : 1) * sizeof(double)));
            // partials, followed by the device copy of the per-slice argument table
            const uint64_t pbytes = XCS_SLICES * pl.n_long * sizeof(double);
            const uint64_t poff = (pbytes + 255) & ~255ull;
            SPRS_TRY_HIP(hipMalloc((void **)&sc.partial, poff + sizeof(SlicedArgs)));
            // a slice without entries launches no tile, so nobody ever writes its partials: they must read as
            // zero (fresh hipMalloc memory usually does, recycled memory does not)
            SPRS_TRY_HIP(hipMemset(sc.partial, 0, poff));
            SlicedArgs sa;
            for (int s = 0; s < XCS_SLICES; ++s) {
                const CsrPiece &sl = pl.slice[s];
                sa.p[s] = TileArgs{sl.indptr, sl.indices, sl.data, sl.pos, sl.tile_row, sc.carry_slices + pl.slice_tile_off[s],
                                   sc.partial + (uint64_t)s * pl.n_long, sl.nnz, sl.ntiles};
            }
            SPRS_TRY_HIP(hipMemcpy((uint8_t *)sc.partial + poff, &sa, sizeof sa, hipMemcpyHostToDevice));
        }
        it = pl.scratch.emplace((void *)stream, sc).first;
    }
    *out = &it->second;
    return SPRS_HIP_OK;
}

// CIDX: column-id type of the pieces the kernels read (the handle's, or uint32 for plan copies)
template <typename CIDX, typename PTR, int TILE>
static int32_t launch_pieces(sprs_hip_csmat *a, SpmvScratch *sc, const double *x, double *y, bool acc,
                             hipStream_t stream) {
    const SpmvPlan &pl = a->plan;
    if (pl.perm) {   // x in the plan's labelling (every column is written: perm is a bijection)
        hipLaunchKernelGGL(rl_permute_x_kernel, dim3((unsigned)((pl.cols + 255) / 256)), dim3(256), 0, stream, x, pl.perm,
                           pl.cols, sc->xp);
        SPRS_TRY_HIP(hipGetLastError());
        x = sc->xp;
    }
    const uint64_t xmask = (uint64_t)options().spmv_xmask;
    // dynamic LDS requested on top of the static arrays only limits how many workgroups share a CU
    const unsigned lds_pad = (unsigned)options().spmv_lds_pad;
    // piece 1: the whole matrix, or its short rows
    if (pl.main.ntiles) {
        const TileArgs ta{pl.main.indptr, pl.main.indices, pl.main.data, pl.main.pos, pl.main.tile_row, sc->carry_main, y,
                          pl.main.nnz,    pl.main.ntiles};
        const dim3 grid((unsigned)pl.main.ntiles), block(BLOCK);
        if (acc) hipLaunchKernelGGL((spmv_tile_kernel<CIDX, PTR, true, TILE>), grid, block, lds_pad, stream, ta, x, xmask);
        else hipLaunchKernelGGL((spmv_tile_kernel<CIDX, PTR, false, TILE>), grid, block, lds_pad, stream, ta, x, xmask);
        SPRS_TRY_HIP(hipGetLastError());
        if (pl.main.ntiles > 1) {
            hipLaunchKernelGGL(spmv_carry_kernel<PTR>, dim3((unsigned)((pl.main.ntiles + 255) / 256)), dim3(256), 0,
                               stream, ta, (uint32_t)TILE);
            SPRS_TRY_HIP(hipGetLastError());
        }
    } else if (!acc) {
        SPRS_TRY_HIP(hipMemsetAsync(y, 0, a->rows * sizeof(double), stream));   // only long rows will be written
    }
    if (!pl.xcs) return SPRS_HIP_OK;

    // piece 2: the long rows, slice s on XCD s
    uint64_t max_tiles = 0;
    for (int s = 0; s < XCS_SLICES; ++s)
        if (pl.slice[s].ntiles > max_tiles) max_tiles = pl.slice[s].ntiles;
    const uint64_t pbytes = XCS_SLICES * pl.n_long * sizeof(double);
    const SlicedArgs *sa = (const SlicedArgs *)((uint8_t *)sc->partial + ((pbytes + 255) & ~255ull));
    if (max_tiles) {
        hipLaunchKernelGGL((spmv_sliced_kernel<CIDX, TILE>), dim3((unsigned)(max_tiles * XCS_SLICES)), dim3(BLOCK),
                           lds_pad, stream, sa, x, xmask);
        SPRS_TRY_HIP(hipGetLastError());
        if (max_tiles > 1) {
            hipLaunchKernelGGL(spmv_sliced_carry_kernel, dim3((unsigned)((max_tiles + 255) / 256), XCS_SLICES),
                               dim3(256), 0, stream, sa, (uint32_t)TILE);
            SPRS_TRY_HIP(hipGetLastError());
        }
    }
    const dim3 rg((unsigned)((pl.n_long + 255) / 256)), rb(256);
    if (acc) hipLaunchKernelGGL(xcs_reduce_kernel<true>, rg, rb, 0, stream, sc->partial, pl.long_rows, y, pl.n_long);
    else hipLaunchKernelGGL(xcs_reduce_kernel<false>, rg, rb, 0, stream, sc->partial, pl.long_rows, y, pl.n_long);
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR>
static int32_t launch_tiled(sprs_hip_csmat *a, const double *x, double *y, bool acc, hipStream_t stream) {
    const Options &o = options();
    SpmvScratch *sc = nullptr;
    // The handle's lock is held until the kernels are launched: another host thread that changes an option (plan rebuild),
    // refreshes or frees the handle cannot pull the plan away between its look-up and the launches that read it.
    std::lock_guard<std::recursive_mutex> lock(a->mu);
    {
        SpmvPlan &pl = a->plan;
        const bool defer = o.spmv_plan_defer != 0 && a->spmv_calls == 0 && !a->prepared;
        if (!pl.built || pl.opt_sig != plan_signature(o) || (pl.light && !defer)) {
#ifndef SPRS_HIP_EMU
            // a plan build allocates, copies to the host and synchronises: none of that may happen on a stream that is being
            // captured into a hipGraph (the graph would replay the build, or the capture would be invalidated half-way)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
                SPRS_FAIL(SPRS_HIP_INVALID_ARG, "SpMV on a capturing stream needs the handle's plan to exist: call sprs_hip_csmat_prepare first");
            (void)hipGetLastError();
#endif
            SPRS_TRY((build_plan<IDX, PTR>(a, stream, defer)));
        }
        ++a->spmv_calls;
        if (!pl.band) SPRS_TRY(get_scratch(pl, stream, &sc));
    }
    const SpmvPlan &pl = a->plan;
    if (pl.band) return band_spmv(a, pl.band, x, y, acc, stream);
    const bool small_idx = pl.idx_bytes == 4;
    if (pl.tile == 2048) {
        if (small_idx) return launch_pieces<uint32_t, PTR, 2048>(a, sc, x, y, acc, stream);
        return launch_pieces<uint64_t, PTR, 2048>(a, sc, x, y, acc, stream);
    }
    if (small_idx) return launch_pieces<uint32_t, PTR, 4096>(a, sc, x, y, acc, stream);
    return launch_pieces<uint64_t, PTR, 4096>(a, sc, x, y, acc, stream);
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
    if (options().spmv_kernel == 2) return launch_rowwave<IDX, PTR>(a, x, y, acc, stream);
    return launch_tiled<IDX, PTR>(a, x, y, acc, stream);
}

int32_t spmv_prepare(sprs_hip_csmat *a, hipStream_t stream) {
    std::lock_guard<std::recursive_mutex> lock(a->mu);
    a->prepared = true;
    if (a->rows == 0 || a->nnz == 0 || options().spmv_kernel == 2) return SPRS_HIP_OK;
    if (a->plan.built && !a->plan.light && a->plan.opt_sig == plan_signature(options())) return SPRS_HIP_OK;
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return build_plan<uint64_t, uint64_t>(a, stream);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return build_plan<uint32_t, uint64_t>(a, stream);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return build_plan<uint64_t, uint32_t>(a, stream);
    return build_plan<uint32_t, uint32_t>(a, stream);
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
