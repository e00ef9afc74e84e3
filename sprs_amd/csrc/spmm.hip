// CSR x dense (row-major) SpMM for gfx950 — device twin of
//   prod::csr_mulacc_dense_rowmaj   sprs/src/sparse/prod.rs:189-214
// (what `&CsMat * &Array2` dispatches to when the rhs has >= 8 columns, csmat.rs:2002-2016):
//     out[i, :] += a_ik * rhs[k, :]     for every stored a_ik, ascending k.
// First "next" row of SURVEY §8(f): the same CSR stream as SpMV, but every gathered x entry
// becomes a whole ROW of the rhs — k contiguous doubles — so one L2 line access now carries
// k/16 .. 1 full line of useful data instead of 8 bytes of it.
//
// Three decompositions (plan cached in the handle):
//  * DEFAULT, the entry stream (spmm_stream_kernel below): a wave takes a tile of 256 consecutive entries, finds their rows
//    from the indptr values that fall into the tile, and its lane groups walk runs of consecutive entries with 16 rhs rows
//    in flight per lane.  5.8 ms at k = 16 on R-MAT 10M: 58 G rhs rows/s, the rate at which this part serves random
//    128-byte lines (scripts/probes/hbm_patterns.hip).
//  * option spmm_stream = 0 (rounds 1-3; also matrices with 2^32 or more columns): every row is cut into CHUNKS of 512
//    entries, one wave per chunk (lane j of group g takes entries g, g + G, ... of the chunk, groups of KP lanes — KP = k
//    rounded up to a power of two — combined with xor-shuffles), partials of multi-chunk rows added in chunk order by a
//    second kernel: deterministic, equal to the reference up to the summation order.  12.1 ms at k = 16: four dependent
//    round trips per ~32-entry row (9.7 ms even with the whole rhs in L2, profiles/r11c).
//  * option spmm_long_row = L > 0: rows of at most L entries instead belong to a GROUP of KP lanes that walks its rows
//    as a little state machine (32 entries and their rhs rows in flight per turn) and adds the products IN ENTRY ORDER
//    into one accumulator per column — the reference's own order, bit for bit, the accumulate form included.  Measured
//    3.4x slower than the chunks (41 ms at k = 16, profiles/r03f), hence opt-in for callers that need the reference's bits.
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

// ---- the entry stream (default) ---------------------------------------------------------------------------------------
// The chunk kernel above walks ROWS: per row of ~32 entries a wave pays four dependent round trips (chunk -> bounds -> indices
// -> rhs rows) — 9.7 ms at k = 16 even when the whole rhs sits in L2 (profiles/r11c), the gathers themselves add a quarter.
// This kernel walks ENTRIES: a wave takes a TILE of 256 consecutive entries of the CSR arrays (coalesced, unconditional loads
// into LDS), finds the row of every entry from the handful of indptr values that fall into the tile (row starts scattered as
// marks, then a max-scan — the rows of a tile are known after ONE round trip whatever their lengths), and its G = 64 / KP lane
// groups each walk a RUN of 256 / G consecutive entries with 16 rhs rows in flight per lane, adding in entry order.  A row
// that lies inside one run is stored directly (its sum has the reference's own order, prod.rs:203-210); the first and the last
// row piece of every run go through LDS and are joined in run order by the wave; a row that crosses tiles leaves a `lead` /
// `trail` partial per tile, added in tile order by a fix-up kernel.  Deterministic, no float atomics.
constexpr int ST_TILE = 256;                 // entries of a tile (one wave)
constexpr int ST_Q = ST_TILE / WAVE;
constexpr uint32_t ST_NONE = 0xffffffffu;
struct alignas(16) Row4 {
    uint32_t x, y, z, w;
};
struct alignas(16) Val2 {
    double x, y;
};

// tile_row[t] = the row that holds entry t * ST_TILE (the last row starting at or before it); tile_row[ntiles] = rows
template <typename PTR>
__global__ void tile_rows_kernel(const PTR *__restrict__ indptr, uint64_t rows, uint64_t ntiles, uint64_t *__restrict__ tile_row) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) {
        tile_row[t] = rows;
        return;
    }
    const uint64_t target = t * ST_TILE;
    uint64_t lo = 0, hi = rows;                          // indptr[lo] <= target < indptr[hi]
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if ((uint64_t)indptr[mid] <= target) lo = mid;
        else hi = mid;
    }
    tile_row[t] = lo;
}

// Where row c of the rhs sits in the re-laid-out copy (option spmm_relayout): the block of 4096 rows stays, the row's place
// inside the block is scattered by an odd multiplier (a bijection of 0 .. 4095), rotated from block to block.  The hub
// columns of a power-law matrix sit at 0, 2^k, 2^j + 2^k ...: their rows of the rhs, 128 bytes apart times such numbers,
// share a few L2 / fabric channels — the same SpMM call took 5.7 or 7.9 ms depending on where the driver had put the rhs,
// and 5.7 always once the columns were permuted at random (profiles/r11y).
constexpr uint32_t RL_BITS = 12, RL_MASK = (1u << RL_BITS) - 1u;
__device__ __forceinline__ uint32_t relaid_row(uint32_t c) {
    const uint32_t hi = c >> RL_BITS;
    return (hi << RL_BITS) | ((c * 0x9E5u + hi * 0x6A7u) & RL_MASK);
}

// the copy: row c of the rhs (any strides) -> row relaid_row(c) of a row-major array with a pitch of KP doubles (rows never
// straddle a 128-byte line, whatever k is)
template <int KP>
__global__ __launch_bounds__(MM_BLOCK) void spmm_relayout_kernel(const double *__restrict__ rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint64_t rows_rhs,
                                                                 uint32_t k, double *__restrict__ dst) {
    const uint32_t j = threadIdx.x % KP;
    const uint64_t c = ((uint64_t)blockIdx.x * MM_BLOCK + threadIdx.x) / KP;
    if (c >= rows_rhs || j >= k) return;
    __builtin_nontemporal_store(rhs[c * ld_rhs + (uint64_t)j * cs_rhs], dst + (uint64_t)relaid_row((uint32_t)c) * KP + j);
}

template <typename IDX, typename PTR, int KP, bool ACC, bool RELAID>
__global__ __launch_bounds__(MM_BLOCK) void spmm_stream_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                               const double *__restrict__ data, uint64_t rows, uint64_t nnz,
                                                               const uint64_t *__restrict__ tile_row, uint64_t ntiles,
                                                               const double *__restrict__ rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k,
                                                               double *__restrict__ out, uint64_t ld_out, uint64_t cs_out,
                                                               double *__restrict__ carry, uint32_t dbg) {
    constexpr int G = WAVE / KP;                         // lane groups = runs of a tile
    constexpr int L = ST_TILE / G;                       // entries of a run
    constexpr int U = L < 16 ? L : 16;                   // rhs rows in flight per lane
    __shared__ uint32_t ecol_s[MM_WAVES][ST_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t erow_s[MM_WAVES][ST_TILE];
    __shared__ __attribute__((aligned(16))) double eval_s[MM_WAVES][ST_TILE];
    __shared__ double seg_s[MM_WAVES][2 * WAVE];         // first / last row piece of every run: [which * 64 + g * KP + j]
    __shared__ uint32_t segrow_s[MM_WAVES][2 * G];
    const uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const uint32_t j = lane % KP, g = lane / KP;
    const uint64_t t = (uint64_t)blockIdx.x * MM_WAVES + w;
    if (t >= ntiles) return;                             // (whole waves; the kernel has no workgroup barrier)
    uint32_t *ecol = ecol_s[w], *erow = erow_s[w], *segrow = segrow_s[w];
    double *eval = eval_s[w], *seg = seg_s[w];
    const uint64_t e0 = t * ST_TILE;
    const uint32_t nt = (uint32_t)(nnz - e0 < (uint64_t)ST_TILE ? nnz - e0 : (uint64_t)ST_TILE);
    // the entries of the tile (past the end of the matrix: the last entry again, never added)
#pragma unroll
    for (int q = 0; q < ST_Q; ++q) {
        const uint32_t pos = (uint32_t)q * WAVE + lane;
        const uint64_t at = e0 + pos < nnz ? e0 + pos : nnz - 1;
        uint32_t col;
        if (DEVTOOLS && (dbg & 1u)) {                     // (developer A/B, option spmm_debug)
            col = (uint32_t)indices[at];
            eval[pos] = data[at];
        } else {
            col = (uint32_t)__builtin_nontemporal_load(indices + at);            // the entries are used once: the L2s are for the rhs rows
            eval[pos] = __builtin_nontemporal_load(data + at);
        }
        ecol[pos] = RELAID ? relaid_row(col) : col;
        erow[pos] = 0u;
    }
    // rows: every non-empty row that starts inside the tile marks its first entry with its distance from r0.  The walk covers
    // the rows (r0, r1] — tile 0 also the rows before r0, the last tile the rows after the last entry — so every EMPTY row is
    // met by exactly one tile, which gives it the zeros of the operator form (csmat.rs:2004; the accumulate form leaves it alone)
    const uint64_t r0 = tile_row[t], r1 = tile_row[t + 1];
    const uint64_t s0 = (uint64_t)indptr[r0];
    const uint64_t rhi = r1 < rows ? r1 : rows - 1;
    wave_sync_lds();
    for (uint64_t r = (t == 0 ? 0 : r0 + 1) + lane; r <= rhi; r += WAVE) {
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        if (e > s) {
            if (r > r0 && s < e0 + nt) erow[s - e0] = (uint32_t)(r - r0);
        } else if constexpr (!ACC) {
            for (uint32_t c = 0; c < k; ++c) out[r * ld_out + (uint64_t)c * cs_out] = 0.0;
        }
    }
    wave_sync_lds();
    uint32_t run_max = 0;
#pragma unroll
    for (int q = 0; q < ST_Q; ++q) {
        const uint32_t pos = (uint32_t)q * WAVE + lane;
        uint32_t o = wave_incl_max_u32(erow[pos]);
        o = o > run_max ? o : run_max;
        run_max = (uint32_t)__builtin_amdgcn_readlane((int)o, WAVE - 1);
        erow[pos] = o;
    }
    wave_sync_lds();
    const uint32_t row_last = erow[nt - 1];
    // the runs
    const uint32_t jj = j < k ? j : k - 1;               // lanes past the last column load what the last column loads, and store nothing
    const bool col_ok = j < k;
    const double *rbase = rhs + (uint64_t)jj * cs_rhs;
    double *obase = out + r0 * ld_out + (uint64_t)jj * cs_out;
    const uint32_t rb = g * L;
    const uint32_t rn = rb < nt ? (nt - rb < (uint32_t)L ? nt - rb : (uint32_t)L) : 0u;
    uint32_t cur = erow[rb < nt ? rb : 0u], first_row = ST_NONE, last_row = ST_NONE;
    bool first = true;
    double acc = 0.0;
    for (uint32_t b = 0; b < (uint32_t)L; b += U) {
        if (__ballot(b < rn) == 0ull) break;             // wave-uniform (the last tile only)
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double *src = rbase + (uint64_t)ecol[rb + b + u] * ld_rhs;
            x[u] = DEVTOOLS && (dbg & 4u) ? __builtin_nontemporal_load(src) : *src;
        }
#pragma unroll
        for (int u4 = 0; u4 < U; u4 += 4) {              // rows and values of four entries per LDS read
            const Row4 rw4 = *reinterpret_cast<const Row4 *>(erow + rb + b + u4);
            const Val2 va = *reinterpret_cast<const Val2 *>(eval + rb + b + u4);
            const Val2 vb = *reinterpret_cast<const Val2 *>(eval + rb + b + u4 + 2);
            const uint32_t rw[4] = {rw4.x, rw4.y, rw4.z, rw4.w};
            const double vv[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = b + (uint32_t)(u4 + e) < rn;
                if (in && rw[e] != cur) {
                    if (first) {
                        seg[lane] = acc;
                        first_row = cur;
                        first = false;
                    } else if (col_ok) {
                        double *dst = obase + (uint64_t)cur * ld_out;
                        const double val = ACC ? *dst + acc : acc;
                        if (DEVTOOLS && (dbg & 2u)) *dst = val;
                        else __builtin_nontemporal_store(val, dst);
                    }
                    cur = rw[e];
                    acc = 0.0;
                }
                const double sum = acc + vv[e] * x[u4 + e];
                acc = in ? sum : acc;
            }
        }
    }
    if (rn) {
        if (first) {
            seg[lane] = acc;
            first_row = cur;
        } else {
            seg[WAVE + lane] = acc;
            last_row = cur;
        }
    }
    if (j == 0) {
        segrow[2 * g] = first_row;
        segrow[2 * g + 1] = last_row;
    }
    wave_sync_lds();
    // the pieces at the run boundaries, joined in run order by the lanes of the first group (lane = column)
    if (lane < (uint32_t)KP && col_ok) {
        const bool lead = s0 < e0;                                    // row r0 began in an earlier tile
        const bool cont = e0 + nt < nnz && r1 == r0 + row_last;       // the last row goes on in the next tile
        double *cbase = carry + 2 * t * (uint64_t)k;
        bool have = false, head = true;
        uint32_t mrow = 0;
        double msum = 0.0;
        auto emit = [&](bool last) {
            if (head && lead) cbase[j] = msum;                        // (a tile inside one long row: its whole sum is a lead)
            else if (last && cont) cbase[k + j] = msum;
            else {
                double *dst = obase + (uint64_t)mrow * ld_out;
                if constexpr (ACC) *dst = *dst + msum;
                else *dst = msum;
            }
            head = false;
        };
        for (int s = 0; s < 2 * G; ++s) {
            const uint32_t row = segrow[s];
            if (row == ST_NONE) continue;
            const double val = seg[(s & 1) * WAVE + (s >> 1) * KP + (int)j];
            if (have && row == mrow) {
                msum = msum + val;
            } else {
                if (have) emit(false);
                mrow = row;
                msum = val;
                have = true;
            }
        }
        if (have) emit(true);
    }
}

// rows that cross tiles: trail of the tile the row starts in, then the leads of the tiles it goes on in, in tile order.
// One lane group per tile boundary (most rows end in the next tile); a row that goes on for more than FIX_SPAN tiles is
// summed by the whole wave afterwards, one such row at a time (its leads dealt to the groups, joined by xor-shuffles).
constexpr uint64_t FIX_SPAN = 8;
template <typename PTR, int KP, bool ACC>
__global__ __launch_bounds__(MM_BLOCK) void spmm_stream_fixup_kernel(const PTR *__restrict__ indptr, const uint64_t *__restrict__ tile_row,
                                                                     uint64_t ntiles, const double *__restrict__ carry, uint32_t k,
                                                                     double *__restrict__ out, uint64_t ld_out, uint64_t cs_out) {
    constexpr int G = WAVE / KP;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t j = lane % KP, g = lane / KP;
    const uint32_t jj = j < k ? j : k - 1;
    const uint64_t t = ((uint64_t)blockIdx.x * MM_BLOCK + threadIdx.x) / KP;
    uint64_t r = 0, tb = 0;
    bool work = false;
    if (t + 1 < ntiles) {                                // (the last tile has no trail)
        const uint64_t e0 = t * ST_TILE, e1 = e0 + ST_TILE;
        r = tile_row[t + 1];                             // the row that holds entry e1
        const uint64_t s = (uint64_t)indptr[r];
        work = s >= e0 && s < e1;                        // it starts in this tile
        if (work) tb = ((uint64_t)indptr[r + 1] - 1) / ST_TILE;       // its last tile
    }
    const bool is_long = work && tb - t > FIX_SPAN;
    if (work && !is_long) {
        double sum = carry[(2 * t + 1) * (uint64_t)k + jj];
        for (uint64_t u = t + 1; u <= tb; ++u) sum = sum + carry[2 * u * (uint64_t)k + jj];
        if (j < k) {
            double *dst = out + r * ld_out + (uint64_t)j * cs_out;
            if constexpr (ACC) *dst = *dst + sum;
            else *dst = sum;
        }
    }
    uint64_t todo = __ballot(is_long && j == 0);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint64_t lt = __shfl(t, src, WAVE), ltb = __shfl(tb, src, WAVE), lr = __shfl(r, src, WAVE);
        double sum = 0.0;
        for (uint64_t u = lt + 1 + g; u <= ltb; u += G) sum = sum + carry[2 * u * (uint64_t)k + jj];
#pragma unroll
        for (int o = KP; o < WAVE; o <<= 1) sum += __shfl_xor(sum, o, WAVE);
        if (g == 0 && j < k) {
            sum = carry[(2 * lt + 1) * (uint64_t)k + j] + sum;
            double *dst = out + lr * ld_out + (uint64_t)j * cs_out;
            if constexpr (ACC) *dst = *dst + sum;
            else *dst = sum;
        }
    }
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
int32_t build_spmm_plan(sprs_hip_csmat *a, bool stream_mode, hipStream_t stream) {
    SpmmPlan &pl = a->mm;
    pl.release();
    pl.long_row = options().spmm_long_row >= 0 ? (uint64_t)options().spmm_long_row : LONG_ROW;
    pl.stream = stream_mode;
    const uint64_t rows = a->rows;
    if (stream_mode) {
        pl.ntiles = (a->nnz + ST_TILE - 1) / ST_TILE;
        SPRS_TRY_HIP(hipMalloc((void **)&pl.tile_row, (pl.ntiles + 1) * 8));
        hipLaunchKernelGGL(tile_rows_kernel<PTR>, dim3((unsigned)((pl.ntiles + 256) / 256)), dim3(256), 0, stream, (const PTR *)a->indptr,
                           rows, pl.ntiles, pl.tile_row);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY_HIP(hipStreamSynchronize(stream));
        pl.built = true;
        return SPRS_HIP_OK;
    }
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

// rows of at most long_row entries (the empty ones among them): the lane-group kernel
template <typename IDX, typename PTR, int KP>
int32_t launch_short_rows(sprs_hip_csmat *a, const double *rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k, double *out, uint64_t ld_out,
                          uint64_t cs_out, bool acc, uint64_t long_row, hipStream_t stream) {
    // groups of KP lanes, group-stride; enough workgroups to fill the chip eight deep
    constexpr uint64_t groups_per_block = MM_BLOCK / KP;
    uint64_t blocks = (a->rows + groups_per_block - 1) / groups_per_block;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const dim3 grid((unsigned)blocks), block(MM_BLOCK);
    if (acc)
        hipLaunchKernelGGL((spmm_rows_kernel<IDX, PTR, KP, true>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, a->rows, rhs, ld_rhs, cs_rhs, k, out, ld_out, cs_out, long_row);
    else
        hipLaunchKernelGGL((spmm_rows_kernel<IDX, PTR, KP, false>), grid, block, 0, stream, (const PTR *)a->indptr,
                           (const IDX *)a->indices, a->data, a->rows, rhs, ld_rhs, cs_rhs, k, out, ld_out, cs_out, long_row);
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR, int KP>
int32_t launch_stream(sprs_hip_csmat *a, const double *rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k, double *out, uint64_t ld_out,
                      uint64_t cs_out, bool acc, double *carry, double *relaid, hipStream_t stream) {
    const SpmmPlan &pl = a->mm;
    const dim3 grid((unsigned)((pl.ntiles + MM_WAVES - 1) / MM_WAVES)), block(MM_BLOCK);
    const dim3 fgrid((unsigned)((pl.ntiles * KP + MM_BLOCK - 1) / MM_BLOCK));
    const uint32_t dbg = (uint32_t)options().spmm_debug;
    if (relaid) {
        // the rhs once through the chip into a compact row-major copy with scattered rows (2 x cols x k x 8 bytes of traffic)
        hipLaunchKernelGGL((spmm_relayout_kernel<KP>), dim3((unsigned)((a->cols * KP + MM_BLOCK - 1) / MM_BLOCK)), block, 0, stream, rhs, ld_rhs,
                           cs_rhs, a->cols, k, relaid);
        rhs = relaid;
        ld_rhs = KP;
        cs_rhs = 1;
    }
#define SPRS_STREAM(ACCV, RLV)                                                                                                         \
    hipLaunchKernelGGL((spmm_stream_kernel<IDX, PTR, KP, ACCV, RLV>), grid, block, 0, stream, (const PTR *)a->indptr,                   \
                       (const IDX *)a->indices, a->data, a->rows, a->nnz, pl.tile_row, pl.ntiles, rhs, ld_rhs, cs_rhs, k, out, ld_out, \
                       cs_out, carry, dbg)
    if (acc) {
        if (relaid) SPRS_STREAM(true, true);
        else SPRS_STREAM(true, false);
        hipLaunchKernelGGL((spmm_stream_fixup_kernel<PTR, KP, true>), fgrid, block, 0, stream, (const PTR *)a->indptr, pl.tile_row, pl.ntiles,
                           carry, k, out, ld_out, cs_out);
    } else {
        if (relaid) SPRS_STREAM(false, true);
        else SPRS_STREAM(false, false);
        hipLaunchKernelGGL((spmm_stream_fixup_kernel<PTR, KP, false>), fgrid, block, 0, stream, (const PTR *)a->indptr, pl.tile_row, pl.ntiles,
                           carry, k, out, ld_out, cs_out);
    }
#undef SPRS_STREAM
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

template <typename IDX, typename PTR, int KP>
int32_t launch_block(sprs_hip_csmat *a, const double *rhs, uint64_t ld_rhs, uint64_t cs_rhs, uint32_t k, double *out, uint64_t ld_out,
                     uint64_t cs_out, bool acc, double *partial, hipStream_t stream) {
    const SpmmPlan &pl = a->mm;
    SPRS_TRY((launch_short_rows<IDX, PTR, KP>(a, rhs, ld_rhs, cs_rhs, k, out, ld_out, cs_out, acc, pl.long_row, stream)));
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
    double *partial = nullptr, *relaid = nullptr;
    std::lock_guard<std::recursive_mutex> lock(a->mu);   // held until the kernels that read the plan are launched
    const uint64_t want_long = options().spmm_long_row >= 0 ? (uint64_t)options().spmm_long_row : LONG_ROW;
    // the entry stream keeps 32-bit column ids in LDS; the lane-group mode (reference bits for short rows) has its own kernels
    // (and 32-bit row distances inside a tile)
    const bool stream_mode = options().spmm_stream != 0 && want_long == 0 && a->cols <= 0xffffffffull && a->rows <= 0xffffffffull && a->nnz != 0;
    // a hypersparse matrix in the operator form: the tile that meets a run of empty rows zeroes them one wave per run, column by
    // column — with many more rows than entries that is a cliff.  The result is cleared in one piece instead and
    // the accumulate kernels run on it: +0.0 + products, the same bits as starting from zero.
    if (!acc && stream_mode && a->rows > 4 * a->nnz + 4096 && k && (cs_out == 1 || ld_out == 1)) {
        if (cs_out == 1 && ld_out == k) SPRS_TRY_HIP(hipMemsetAsync(out, 0, a->rows * k * sizeof(double), stream));
        else if (ld_out == 1 && cs_out == a->rows) SPRS_TRY_HIP(hipMemsetAsync(out, 0, a->rows * k * sizeof(double), stream));
        else if (cs_out == 1) SPRS_TRY_HIP(hipMemset2DAsync(out, ld_out * sizeof(double), 0, k * sizeof(double), a->rows, stream));
        else SPRS_TRY_HIP(hipMemset2DAsync(out, cs_out * sizeof(double), 0, a->rows * sizeof(double), k, stream));
        acc = true;
    }
    {
        if (!a->mm.built || a->mm.long_row != want_long || a->mm.stream != stream_mode) SPRS_TRY(build_spmm_plan<PTR>(a, stream_mode, stream));
        SpmmPlan &pl = a->mm;
        const uint64_t kb = k < 64 ? k : 64;
        const uint64_t need = (stream_mode ? 2 * pl.ntiles : pl.n_multi ? pl.nchunks : 0) * kb * sizeof(double);
        auto &slot = pl.partial[(void *)stream];
        if (slot.second < need) {
            if (slot.first) (void)hipFree(slot.first);
            slot.first = nullptr;
            slot.second = 0;
            SPRS_TRY_HIP(hipMalloc((void **)&slot.first, need));
            slot.second = need;
        }
        partial = slot.first;
        // the re-laid-out copy of the rhs (option spmm_relayout; auto: a rhs that is not row-major — a column of it is a separate
        // line per entry — or one of 256 MiB and more per column block with 24 and more gathers per row to pay for the copy:
        // R-MAT 4M x 32 2.55 -> 2.20 ms at k = 16, 1M x 16 0.32 -> 0.38 ms, profiles/r11zc)
        // ... and whose row pitch invites the trouble: a power of two (hub ids 2^j times a pitch of 2^m bytes; a pitch of 192 or 384
        // bytes — k = 24, 48 — spreads the hub rows by itself: 10.7 ms in place against 15.5 ms at k = 32, and the copy only costs
        // there, profiles/r12b), or rows of fewer than 16 columns that straddle lines (k = 12: 96-byte rows; the copy's pitch is 128)
        const int64_t rl = options().spmm_relayout;
        const bool big = a->cols * kb * sizeof(double) >= (256ull << 20) && a->nnz >= 24 * a->cols;
        const bool pitch = (ld_rhs & (ld_rhs - 1)) == 0 || (kb < 16 && (ld_rhs * sizeof(double)) % 128 != 0);
        const bool want = stream_mode && (rl == 1 || (rl == 0 && (cs_rhs != 1 || (big && pitch))));
        if (want) {
            const uint64_t rows_pad = (a->cols + RL_MASK) & ~(uint64_t)RL_MASK;
            const uint64_t kp = kb <= 8 ? 8 : kb <= 16 ? 16 : kb <= 32 ? 32 : 64;
            const uint64_t bytes = rows_pad * kp * sizeof(double);
            auto &rs = pl.relaid[(void *)stream];
            if (rs.second < bytes) {
                if (rs.first) (void)hipFree(rs.first);
                rs.first = nullptr;
                rs.second = 0;
                // (no room for a second copy of the rhs: the kernel gathers from the caller's, as with spmm_relayout = 2)
                if (hipMalloc((void **)&rs.first, bytes) == hipSuccess) rs.second = bytes;
                else {
                    rs.first = nullptr;
                    (void)hipGetLastError();
                }
            }
            relaid = rs.first;
        }
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
        if (stream_mode) {
            if (kb <= 8) st = launch_stream<IDX, PTR, 8>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, relaid, stream);
            else if (kb <= 16) st = launch_stream<IDX, PTR, 16>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, relaid, stream);
            else if (kb <= 32) st = launch_stream<IDX, PTR, 32>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, relaid, stream);
            else st = launch_stream<IDX, PTR, 64>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, relaid, stream);
        } else if (kb <= 8) st = launch_block<IDX, PTR, 8>(a, r, ld_rhs, cs_rhs, kb, o, ld_out, cs_out, acc, partial, stream);
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
    drop(tile_row);
    first_chunk = chunk_row = multi_rows = tile_row = nullptr;
    ntiles = 0;
    for (auto &kv : partial) drop(kv.second.first);
    partial.clear();
    for (auto &kv : relaid) drop(kv.second.first);
    relaid.clear();
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
