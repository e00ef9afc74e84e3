// BANDED SpMV PLAN for power-law matrices whose x does not fit the L2s (BASELINE config 4, R-MAT 10M)
// — device twin of prod::mul_acc_mat_vec_csr (sprs/src/sparse/prod.rs:103-127) and of the one-column
// prod::csr_mulacc_dense_colmaj (prod.rs:274-298), like spmv.hip.
//
// Why: with every x[col] gathered through L1/L2 the kernel moves one 128-byte line from L2 to L1 per
// 8 useful bytes (0.87 L2 requests per non-zero on R-MAT 10M, profiles/r01zr...): it is bound by the
// L2 -> L1 fill bandwidth, not by HBM, and stops at 38 % of the HBM roofline.  On a power-law matrix a
// small set of columns carries most of the entries (R-MAT 10M: the 2e5 most referenced columns of 1e7
// hold ~2/3 of the non-zeros), and the plan may lay the matrix out as it likes.  So:
//
//   * columns are relabelled by popularity class (rl_* kernels, spmv_shared.hpp), x is permuted into
//     that order once per SpMV (xp);
//   * HOT BAND: labels [0, NH * 8192).  Hot slice k = the entries of the long rows (>= split entries)
//     whose label lies in [8192 k, 8192 (k+1)), stored as a CSR piece over the rows that HAVE entries
//     there, row after row: 8-byte value + 16-BIT local column id (10 B per entry instead of 16).  The
//     hot kernel keeps the 8192 x entries of its slice in LDS (64 KiB, loaded coalesced) and gathers
//     from LDS: no L1/L2 traffic per entry at all, the slice streams at HBM speed;
//   * COLD REST: the other entries of the long rows, in 8 pieces by a hash of their x line (piece s runs
//     on XCD s, as in the XCD-sliced plan of spmv.hip), optionally in several label ranges ("phases")
//     so that a piece's x window fits one 4 MiB L2;
//   * SHORT ROWS: one CSR piece over the rows that are not empty (47 % of the rows of R-MAT 10M are),
//     so that the boundary walk no longer visits empty rows.
//   Every piece is processed nnz-tile by nnz-tile (coalesced, balanced whatever the row lengths),
//   products staged in LDS, row segments summed by one lane / one wave each, multi-tile rows fixed up
//   through carries — the machinery of spmv.hip.  A piece writes one partial sum per (row, piece) into
//   its slab; rows without entries in the piece keep the zero the slab was created with.  A last kernel
//   adds the slabs of a long row in piece order.  No float atomics: bit-reproducible run to run.
#include "spmv_shared.hpp"

#include <cstdlib>
#include <vector>

namespace sprs_hip {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int CB_LOG2 = 13;
constexpr int CB = 1 << CB_LOG2;     // labels per hot slice = doubles of the x tile in LDS (64 KiB)
constexpr int HT = 8192;             // entries per hot tile
constexpr int HNT = 1024;            // threads of a hot workgroup (16 waves, one workgroup per CU)
constexpr int CT = 4096;             // entries per cold / short tile
constexpr int CNT = 256;             // threads of a cold workgroup (3 per CU)
constexpr int CPASS = CT / (CNT * 2);
constexpr int MAX_HOT = 96;
constexpr int MAX_PHASES = 8;
constexpr int MAX_PIECES = MAX_HOT + 8 * MAX_PHASES + 1;

// One CSR piece of the plan, as the kernels see it.
struct BandPiece {
    const uint32_t *ptr;        // nr + 1 entry offsets, relative to the piece
    const uint32_t *rowidx;     // nr: where the sum of compact row r goes in `out`
    const uint32_t *tile_row;   // ntiles + 1: first compact row starting at / after tile c
    double *carry;              // ntiles
    double *out;                // slab of partial sums (n_long doubles); the short piece writes y instead
    uint64_t ent0;              // first entry of the piece in the value / column-id arrays of its class
    uint64_t nnz;
    uint32_t nr, ntiles;
    uint32_t x0;                // hot: first label of the slice
    uint32_t to_y;              // short piece
};

struct ColdGroup {              // a run of blocks of the cold launch
    uint32_t first_block, first_piece, npieces;   // npieces 8: block b -> piece b % 8 (XCD b % 8), tile b / 8; 1: tile b
};

template <int T, int SEGC>
struct SegLds {
    uint32_t segb[SEGC + 1];        // tile-local boundaries of the staged segments
    uint32_t segr[SEGC + 1];        // where the sum of a staged segment goes (index into the piece's output)
    uint32_t longlist[T / LONG_SEG + 1];
    uint32_t nlong;
};

template <bool LDS_ONLY>
__device__ __forceinline__ void tile_barrier() {
    if constexpr (LDS_ONLY) lds_barrier();   // does not wait for global loads in flight (the prefetch of the next tile)
    else __syncthreads();
}

// Row-segment sums of one tile whose products are in prod[0 .. cnt): segment 0 = head (the tail of a row
// that started in an earlier tile), segment j >= 1 = compact row R0 + j - 1.  Segments are staged SEGC at a
// time: stage(j0, n) fills L.segb[0 .. n] (boundaries) and L.segr[0 .. n) (output index of segment j0 + t);
// emit(j, out_index, sum) is called once per segment.
template <int NT, int T, int SEGC, bool LDS_ONLY, typename Stage, typename Emit>
__device__ __forceinline__ void segment_sums(const double *prod, SegLds<T, SEGC> &L, uint32_t S, Stage stage, Emit emit) {
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    if (tid == 0) L.nlong = 0;
    for (uint32_t j0 = 0; j0 < S; j0 += SEGC) {
        const uint32_t n = S - j0 < (uint32_t)SEGC ? S - j0 : (uint32_t)SEGC;
        stage(j0, n);
        tile_barrier<LDS_ONLY>();   // prod[], segb[], segr[], nlong visible
        for (uint32_t t = tid; t < n; t += NT) {
            const uint32_t sa = L.segb[t], sb = L.segb[t + 1];
            if (sb - sa >= LONG_SEG) {
                L.longlist[atomicAdd(&L.nlong, 1u)] = t;
            } else {
                double s = 0.0;
                for (uint32_t k = sa; k < sb; ++k) s += prod[k];
                emit(j0 + t, L.segr[t], s);
            }
        }
        tile_barrier<LDS_ONLY>();   // longlist complete
        const uint32_t nl = L.nlong;
        for (uint32_t q = wave; q < nl; q += NT / WAVE) {
            const uint32_t t = L.longlist[q];
            const uint32_t sa = L.segb[t], sb = L.segb[t + 1];
            double s = 0.0;
            for (uint32_t k = sa + lane; k < sb; k += WAVE) s += prod[k];
            s = wave_sum(s);
            if (lane == 0) emit(j0 + t, L.segr[t], s);
        }
        tile_barrier<LDS_ONLY>();   // everyone done with segb / longlist / prod
        if (tid == 0) L.nlong = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// hot slices: x tile in LDS, 16-bit column ids, row sums in registers.
//
// A workgroup (16 waves) loads the 8192 x entries of ONE slice into LDS once and then takes G consecutive
// blocks of 8192 entries of that slice.  Inside a block every WAVE owns a WAVE TILE of 512 entries and
// runs on its own: no workgroup barrier after the x tile is in place.
// Layout of wave tile w (the plan owns it, so it is whatever the kernel reads best):
//   * lane l works on the 8 CONSECUTIVE entries 8 l .. 8 l + 7 of the tile;
//   * column ids in natural order: cid[512 w + i], one 16-byte load per lane; 13 bits of local column,
//     bit 15 = "this entry is the first of its row in this slice" (the row structure travels with the stream:
//     the kernel reads no row offsets at all);
//   * values transposed so that the four coalesced 16-byte loads of a lane return exactly its entries:
//     entry 8 l + 2 p + e is stored at vals[512 w + 128 p + 2 l + e];
//   * both arrays are padded with zeros to whole blocks; the padding belongs to no row.
// Row sums: every lane folds its 8 products serially (runs that start AND end inside the lane are written
// at once), then one segmented scan over the 64 lanes (shuffles) completes the runs that cross lanes; the
// run open at the start of the tile goes to carry[w] (band_carry_kernel adds it to the row that owns it), the
// run open at its end is the partial sum of the last row starting in the tile.  Where a sum goes comes from
// rowidx[tile_row[w] + ordinal of the row inside the tile], prefetched a tile ahead and parked in LDS.
// History (profiles/r02a, r02b): one 8192-entry tile per workgroup iteration with products staged in LDS and
// three workgroup barriers: 2.9 TB/s (one workgroup per CU, nothing overlaps); the same per wave: 3.1 TB/s
// (bound by the serial LDS read-add loops of the segment sums).
// ---------------------------------------------------------------------------------------------
constexpr int WT = 512;                       // entries per wave tile
constexpr int WPB = HT / WT;                  // wave tiles per block = waves per workgroup
constexpr int WPASS = WT / (WAVE * 2);        // 16-byte value loads per lane and tile
constexpr int EPL = WT / WAVE;                // entries per lane
constexpr int NPRE = 4;                       // output indices prefetched per lane (rows 64 q + lane of the tile)
constexpr uint32_t ROW_START = 0x8000u;       // flag bit in a hot column id
static_assert(WPB * WAVE == HNT && WPASS == 4 && EPL == 8, "hot kernel geometry");

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave complete in order; this keeps the COMPILER from moving them across the hand-over
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(HNT) void band_hot_kernel(const BandPiece *__restrict__ pieces,
                                                       const uint32_t *__restrict__ wg_off, uint32_t nh, uint32_t G,
                                                       const double *__restrict__ vals, const uint16_t *__restrict__ cid,
                                                       const double *__restrict__ xp) {
    __shared__ __attribute__((aligned(16))) double xs[CB];
    __shared__ uint32_t ridx_s[WPB][NPRE * WAVE];
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t k = 0;
    while (k + 1 < nh && blockIdx.x >= wg_off[k + 1]) ++k;       // block-uniform
    const BandPiece d = pieces[k];
    const uint32_t nwt = d.ntiles;                               // wave tiles of the slice
    const uint32_t b0 = (blockIdx.x - wg_off[k]) * G;            // first block of this workgroup
    // x tile of the slice -> LDS (xp is padded to a whole number of tiles)
    {
        const dbl2 *src = (const dbl2 *)(xp + d.x0);
        dbl2 v[CB / (2 * HNT)];
#pragma unroll
        for (int q = 0; q < CB / (2 * HNT); ++q) v[q] = src[q * HNT + tid];
#pragma unroll
        for (int q = 0; q < CB / (2 * HNT); ++q) *(dbl2 *)&xs[2 * (q * HNT + tid)] = v[q];
    }
    uint32_t *ridx = ridx_s[wave];

    // pipeline registers: the next tile of this wave (stream + output indices in flight), tile_row of the one after
    dbl2 av[WPASS];
    u32x4 cw;
    uint32_t po[NPRE];
    uint32_t R0n = 0, R0nn = 0;
    auto tile_of = [&](uint32_t it) { return (b0 + it) * (uint32_t)WPB + wave; };
    auto request_row = [&](uint32_t w) { return d.tile_row[w < nwt ? w : nwt - 1]; };   // past the slice: harmless reload
    auto request_tile = [&](uint32_t w, uint32_t r0) {
        const uint64_t g = d.ent0 + (uint64_t)(w < nwt ? w : nwt - 1) * WT;
#pragma unroll
        for (int p = 0; p < WPASS; ++p) av[p] = __builtin_nontemporal_load((const dbl2 *)(vals + g + p * (WAVE * 2) + lane * 2));
        cw = __builtin_nontemporal_load((const u32x4 *)(cid + g + lane * EPL));
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const uint32_t r = r0 + q * WAVE + lane;             // the row that starts at the tile's flag number 64 q + lane
            po[q] = d.rowidx[r < d.nr ? r : d.nr - 1];
        }
    };
    if (tile_of(0) < nwt) {                                      // (nwt > 0 for every launched workgroup)
        R0n = request_row(tile_of(0));
        request_tile(tile_of(0), R0n);
        R0nn = request_row(tile_of(1));
    }
    __syncthreads();   // xs complete; the only workgroup barrier
    for (uint32_t it = 0; it < G; ++it) {
        const uint32_t w = tile_of(it);
        if (w >= nwt) break;                                     // wave-uniform
        const uint64_t base = (uint64_t)w * WT;
        const uint32_t cnt = d.nnz - base < (uint64_t)WT ? (uint32_t)(d.nnz - base) : (uint32_t)WT;
        // ---- products of the lane's 8 consecutive entries, row-start flags ---------------------------
        double pr[EPL];
        uint32_t fb = 0;
#pragma unroll
        for (int p = 0; p < WPASS; ++p) {
            const uint32_t c2 = cw[p];
            const uint32_t i0 = lane * EPL + 2 * p;
            pr[2 * p] = i0 < cnt ? av[p][0] * xs[c2 & (CB - 1)] : 0.0;
            pr[2 * p + 1] = i0 + 1 < cnt ? av[p][1] * xs[(c2 >> 16) & (CB - 1)] : 0.0;
            fb |= ((c2 >> 15) & 1u) << (2 * p);
            fb |= ((c2 >> 31) & 1u) << (2 * p + 1);
        }
        const uint32_t R0 = R0n;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) ridx[q * WAVE + lane] = po[q];
        // ---- requests for the next tile of this wave ----------------------------------------------------
        R0n = R0nn;
        if (it + 1 < G && tile_of(it + 1) < nwt) {
            request_tile(tile_of(it + 1), R0n);
            R0nn = request_row(tile_of(it + 2));
        }
        wave_lds_fence();                                        // ridx[] of this wave is in place
        // ---- rows starting in lower lanes: the ordinal of this lane's first row inside the tile ----------
        uint32_t prefix = 0, nf = 0;
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const unsigned long long m = __ballot((fb >> q) & 1u);
            prefix += (uint32_t)__popcll(m & below);
            nf += (uint32_t)__popcll(m);
        }
        auto out_index = [&](uint32_t j) -> uint32_t {           // where the sum of the tile's j-th starting row goes
            return j < (uint32_t)(NPRE * WAVE) ? ridx[j] : d.rowidx[R0 + j];
        };
        // ---- serial fold of the lane's entries -------------------------------------------------------------
        double run = 0.0, head = 0.0;
        uint32_t seen = 0;                                       // rows started in this lane so far
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            if ((fb >> q) & 1u) {
                if (seen == 0) head = run;                       // the run that was open when the lane began ends here
                else d.out[out_index(prefix + seen - 1)] = run;  // a row that lies inside the lane
                run = 0.0;
                ++seen;
            }
            run += pr[q];
        }
        // ---- segmented scan over the lanes: S = sum of the run that is open at the END of the lane -------------
        double S = run;
        uint32_t F = seen ? 1u : 0u;
#pragma unroll
        for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
            const double vs = __shfl_up(S, dlt, WAVE);
            const uint32_t fs = __shfl_up(F, dlt, WAVE);
            if (lane >= (uint32_t)dlt) {
                if (!F) S = vs + S;
                F |= fs;
            }
        }
        double before = __shfl_up(S, 1, WAVE);                   // open run at the end of the previous lane
        if (lane == 0) before = 0.0;
        if (seen) {
            const double v = before + head;                      // the run that ends at this lane's first row start
            if (prefix == 0) d.carry[w] = v;                     // ... began before the tile
            else d.out[out_index(prefix - 1)] = v;
        }
        if (lane == WAVE - 1) {                                  // the run still open at the end of the tile
            if (nf == 0) d.carry[w] = S;                         // no row starts in the tile: all of it is head
            else d.out[out_index(nf - 1)] = S;                   // partial sum of the last row starting here
        }
        wave_lds_fence();                                        // done with ridx[] before the next tile overwrites it
    }
}

// ---------------------------------------------------------------------------------------------
// cold pieces and the short rows: x gathered through L1 / L2 from xp, 32-bit labels
// ---------------------------------------------------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(CNT) void band_cold_kernel(const BandPiece *__restrict__ pieces,
                                                        const ColdGroup *__restrict__ groups, uint32_t ngroups,
                                                        const double *__restrict__ vals, const uint32_t *__restrict__ cid,
                                                        const double *__restrict__ xp, double *__restrict__ y,
                                                        uint32_t block0) {
    __shared__ __attribute__((aligned(16))) double prod[CT];
    __shared__ SegLds<CT, SEG_CHUNK> L;
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = blockIdx.x + block0;
    uint32_t g = 0;
    while (g + 1 < ngroups && bid >= groups[g + 1].first_block) ++g;     // block-uniform
    const ColdGroup cg = groups[g];
    const uint32_t lb = bid - cg.first_block;
    const uint32_t pi = cg.first_piece + (cg.npieces == 1 ? 0u : (lb & 7u));
    const uint32_t t = cg.npieces == 1 ? lb : (lb >> 3);
    const BandPiece d = pieces[pi];
    if (t >= d.ntiles) return;
    const uint64_t base = (uint64_t)t * CT;
    const uint32_t cnt = d.nnz - base < (uint64_t)CT ? (uint32_t)(d.nnz - base) : (uint32_t)CT;
    const uint64_t lim = base + cnt;
    const double *dp = vals + d.ent0 + base;
    const uint32_t *ip = cid + d.ent0 + base;
    if (cnt == (uint32_t)CT) {
        u32x2 ix[CPASS];
        dbl2 av[CPASS];
#pragma unroll
        for (int p = 0; p < CPASS; ++p) {
            const uint32_t i = p * (CNT * 2) + tid * 2;
            ix[p] = __builtin_nontemporal_load((const u32x2 *)(ip + i));
            av[p] = __builtin_nontemporal_load((const dbl2 *)(dp + i));
        }
        double xv[CPASS][2];
#pragma unroll
        for (int p = 0; p < CPASS; ++p) {
            xv[p][0] = xp[ix[p][0]];
            xv[p][1] = xp[ix[p][1]];
        }
#pragma unroll
        for (int p = 0; p < CPASS; ++p) {
            dbl2 pr;
            pr[0] = av[p][0] * xv[p][0];
            pr[1] = av[p][1] * xv[p][1];
            *(dbl2 *)&prod[p * (CNT * 2) + tid * 2] = pr;
        }
    } else {
#pragma unroll
        for (int p = 0; p < CPASS; ++p) {
            const uint32_t i = p * (CNT * 2) + tid * 2;
            prod[i] = i < cnt ? dp[i] * xp[ip[i]] : 0.0;
            prod[i + 1] = i + 1 < cnt ? dp[i + 1] * xp[ip[i + 1]] : 0.0;
        }
    }
    const uint32_t R0 = d.tile_row[t], R1 = d.tile_row[t + 1];
    double *out = d.to_y ? y : d.out;
    const bool acc = ACC && d.to_y;
    segment_sums<CNT, CT, SEG_CHUNK, false>(
        prod, L, R1 - R0 + 1,
        [&](uint32_t j0, uint32_t n) {
            for (uint32_t u = tid; u <= n; u += CNT) {
                const uint32_t j = j0 + u;
                uint32_t b = 0, o = 0;
                if (j != 0) {
                    const uint32_t r = R0 + j - 1;
                    const uint64_t v = (uint64_t)d.ptr[r];       // r <= nr: ptr has nr + 1 entries
                    b = (uint32_t)((v < lim ? v : lim) - base);
                    if (r < d.nr) o = d.rowidx[r];
                }
                L.segb[u] = b;
                L.segr[u] = o;
            }
        },
        [&](uint32_t j, uint32_t o, double s) {
            if (j == 0) d.carry[t] = s;
            else if (acc) out[o] = out[o] + s;     // every compact row has entries: empty rows are never touched (prod.rs:120-126)
            else out[o] = s;
        });
}

// a row that spans several tiles of a piece gets the heads of the later tiles added in tile order
__global__ void band_carry_kernel(const BandPiece *__restrict__ pieces, uint32_t hot_pieces, double *__restrict__ y) {
    const BandPiece d = pieces[blockIdx.y];
    const uint32_t T = blockIdx.y < hot_pieces ? (uint32_t)WT : (uint32_t)CT;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c + 1 >= d.ntiles) return;
    const uint32_t R0 = d.tile_row[c], R1 = d.tile_row[c + 1];
    if (R1 == R0) return;                                              // no row starts in tile c
    if ((uint64_t)d.ptr[R1] <= (uint64_t)(c + 1) * T) return;          // its last row ends inside tile c
    double acc = 0.0;
    for (uint32_t e = c + 1; e < d.ntiles && d.tile_row[e] == R1; ++e) acc += d.carry[e];
    double *out = d.to_y ? y : d.out;
    out[d.rowidx[R1 - 1]] += acc;
}

// y[long_rows[j]] (+)= sum over the pieces, in piece order
template <bool ACC>
__global__ void band_reduce_kernel(const double *__restrict__ partial, const uint32_t *__restrict__ long_rows,
                                   double *__restrict__ y, uint32_t n_long, uint32_t npieces) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_long) return;
    double s = partial[j];
    for (uint32_t k = 1; k < npieces; ++k) s += partial[(uint64_t)k * n_long + j];
    const uint32_t r = long_rows[j];
    if constexpr (ACC) y[r] = y[r] + s;
    else y[r] = s;
}

// ---------------------------------------------------------------------------------------------
// plan building (one-time, on the device)
// ---------------------------------------------------------------------------------------------
struct SliceMap {       // label -> piece
    uint32_t nh, phases;
    uint64_t hot_labels;      // nh * CB
    uint64_t phase_width;     // labels per phase of the cold rest
};

__device__ __forceinline__ uint32_t piece_of_label(const SliceMap &m, uint64_t label) {
    if (label < m.hot_labels) return (uint32_t)(label >> CB_LOG2);
    uint64_t ph = (label - m.hot_labels) / m.phase_width;
    if (ph >= m.phases) ph = m.phases - 1;
    return m.nh + (uint32_t)ph * 8u + x_slice(label);
}

template <typename PTR>
__global__ void bp_classify_kernel(const PTR *__restrict__ indptr, uint64_t rows, uint64_t split,
                                   uint64_t *__restrict__ short_flag, uint64_t *__restrict__ short_len,
                                   uint64_t *__restrict__ long_flag) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t len = (uint64_t)indptr[r + 1] - (uint64_t)indptr[r];
    const bool is_long = len >= split;
    short_flag[r] = (!is_long && len) ? 1 : 0;
    short_len[r] = is_long ? 0 : len;
    long_flag[r] = is_long ? 1 : 0;
}

// short rows -> their compact CSR piece (labels instead of columns); long rows -> long_rows
template <typename IDX, typename PTR>
__global__ void bp_fill_short_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                     const double *__restrict__ data, uint64_t rows,
                                     const uint64_t *__restrict__ short_pos, const uint64_t *__restrict__ short_ptr,
                                     const uint64_t *__restrict__ long_pos, const uint32_t *__restrict__ perm,
                                     uint32_t *__restrict__ s_rowidx, uint32_t *__restrict__ s_ptr,
                                     uint32_t *__restrict__ s_cid, double *__restrict__ s_val,
                                     uint32_t *__restrict__ long_rows, uint64_t split) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    if (r == rows) {
        s_ptr[short_pos[rows]] = (uint32_t)short_ptr[rows];
        return;
    }
    const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
    if (e == s) return;
    if (e - s >= split) {
        long_rows[long_pos[r]] = (uint32_t)r;
        return;
    }
    const uint64_t i = short_pos[r];
    uint64_t d = short_ptr[r];
    s_rowidx[i] = (uint32_t)r;
    s_ptr[i] = (uint32_t)d;
    for (uint64_t p = s; p < e; ++p, ++d) {
        s_cid[d] = perm[indices[p]];
        s_val[d] = data[p];
    }
}

// entries of long row j per piece: cnt[k * n_long + j] (and 1 where that is not zero)
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void bp_count_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                       const uint32_t *__restrict__ long_rows, uint64_t n_long,
                                                       const uint32_t *__restrict__ perm, SliceMap map, uint32_t npieces,
                                                       uint64_t *__restrict__ cnt, uint64_t *__restrict__ nz) {
    __shared__ uint32_t hist[4][MAX_PIECES];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    for (uint64_t j = w0; j < n_long; j += nw) {
        for (uint32_t k = lane; k < npieces; k += WAVE) hist[wave][k] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint64_t r = long_rows[j];
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        for (uint64_t p = s + lane; p < e; p += WAVE) atomicAdd(&hist[wave][piece_of_label(map, perm[indices[p]])], 1u);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k < npieces; k += WAVE) {
            const uint32_t c = hist[wave][k];
            cnt[(uint64_t)k * n_long + j] = c;
            nz[(uint64_t)k * n_long + j] = c ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void bp_starts_kernel(const uint64_t *__restrict__ pos, const uint64_t *__restrict__ pair, uint64_t n_long,
                                 uint32_t npieces, uint64_t *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > npieces) return;
    out[k] = pos[(uint64_t)k * n_long];
    out[npieces + 1 + k] = pair[(uint64_t)k * n_long];
}

struct PieceBuild {          // host-computed placement of a piece, read by the scatter kernels
    uint64_t start;          // position of the piece in the piece-major concatenation (scan of cnt)
    uint64_t pair0;          // first (row, piece) pair of the piece (scan of nz)
    uint64_t ent0;           // first entry in the class's arrays
    uint64_t nnz;
    uint32_t ptr_off, row_off;   // offsets into ptr_all / rowidx_all
    uint32_t hot, x0;
};

// stable partition of every long row into its pieces (the entries keep their order inside the row)
template <typename IDX, typename PTR>
__global__ __launch_bounds__(256) void bp_scatter_kernel(const PTR *__restrict__ indptr, const IDX *__restrict__ indices,
                                                         const double *__restrict__ data,
                                                         const uint32_t *__restrict__ long_rows, uint64_t n_long,
                                                         const uint32_t *__restrict__ perm, SliceMap map, uint32_t npieces,
                                                         const uint64_t *__restrict__ pos,
                                                         const PieceBuild *__restrict__ pb,
                                                         double *__restrict__ vals_hot, uint16_t *__restrict__ cid_hot,
                                                         double *__restrict__ vals_cold, uint32_t *__restrict__ cid_cold) {
    __shared__ uint32_t fill[4][MAX_PIECES];       // entries of the row already placed, per piece
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    for (uint64_t j = w0; j < n_long; j += nw) {
        for (uint32_t k = lane; k < npieces; k += WAVE) fill[wave][k] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint64_t r = long_rows[j];
        const uint64_t s = (uint64_t)indptr[r], e = (uint64_t)indptr[r + 1];
        for (uint64_t p0 = s; p0 < e; p0 += WAVE) {
            const uint64_t p = p0 + lane;
            const bool valid = p < e;
            const uint32_t label = valid ? perm[indices[p]] : 0u;
            const double v = valid ? data[p] : 0.0;
            const uint32_t k = valid ? piece_of_label(map, label) : 0xFFFFFFFFu;
            // rank of the entry among the lanes of this batch that go to the same piece (lane order = row order)
            uint32_t rank = 0, group = 0;
            unsigned long long todo = __ballot(valid);
            while (todo) {                                  // wave-uniform: one round per distinct piece of the batch
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t kk = (uint32_t)__shfl((int)k, leader, WAVE);
                const unsigned long long m = __ballot(valid && k == kk);
                if (valid && k == kk) {
                    rank = (uint32_t)__popcll(m & below);
                    group = (uint32_t)__popcll(m);
                }
                todo &= ~m;
            }
            uint32_t before = 0;
            if (valid) before = fill[wave][k];
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) fill[wave][k] = before + group;   // one writer per piece
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const PieceBuild b = pb[k];
                const uint64_t e_rel = pos[(uint64_t)k * n_long + j] - b.start + before + rank;   // entry number inside the piece
                if (b.hot) {
                    const uint64_t tile = e_rel / WT;              // wave tile; entry i = 8 l + 2 p + e of it belongs to lane l
                    const uint32_t i = (uint32_t)(e_rel % WT);
                    const uint32_t ll = i / EPL, pp = (i % EPL) / 2, ee = i & 1u;
                    vals_hot[b.ent0 + tile * WT + pp * (WAVE * 2) + ll * 2 + ee] = v;
                    cid_hot[b.ent0 + e_rel] = (uint16_t)((label - b.x0) | (before + rank == 0 ? ROW_START : 0u));
                } else {
                    vals_cold[b.ent0 + e_rel] = v;
                    cid_cold[b.ent0 + e_rel] = label;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// compact row lists of the pieces: one thread per (piece, long row)
__global__ void bp_rows_kernel(const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ pos,
                               const uint64_t *__restrict__ pair, uint64_t n_long, uint32_t npieces,
                               const PieceBuild *__restrict__ pb, uint32_t *__restrict__ ptr_all,
                               uint32_t *__restrict__ rowidx_all) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)npieces * n_long) return;
    const uint32_t k = (uint32_t)(idx / n_long);
    const uint64_t j = idx - (uint64_t)k * n_long;
    const PieceBuild b = pb[k];
    if (j == 0) ptr_all[b.ptr_off + (pair[idx + n_long] - b.pair0)] = (uint32_t)b.nnz;    // end of the last row
    if (cnt[idx]) {
        const uint64_t q = pair[idx] - b.pair0;
        rowidx_all[b.row_off + q] = (uint32_t)j;
        ptr_all[b.ptr_off + q] = (uint32_t)(pos[idx] - b.start);
    }
}

struct TileRowJob {
    const uint32_t *ptr;
    uint32_t *tile_row;
    uint32_t nr, ntiles, T;
};

__global__ void bp_tile_rows_kernel(const TileRowJob *__restrict__ jobs) {
    const TileRowJob jb = jobs[blockIdx.y];
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > jb.ntiles) return;
    if (c == jb.ntiles) {
        jb.tile_row[c] = jb.nr;
        return;
    }
    const uint64_t target = (uint64_t)c * jb.T;
    uint32_t lo = 0, hi = jb.nr;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)jb.ptr[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    jb.tile_row[c] = lo;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct BandScratch {        // per stream
    double *partial = nullptr, *carry = nullptr, *xp = nullptr;
    BandPiece *pieces = nullptr;
};

struct BandPlan {
    uint32_t nh = 0, phases = 1, npieces = 0;      // npieces = nh + 8 phases (the short piece comes after them)
    uint32_t n_long = 0, n_short_rows = 0, G = 4;
    uint64_t cols = 0, cols_pad = 0;
    uint32_t *perm = nullptr, *long_rows = nullptr;
    double *vals_hot = nullptr, *vals_cold = nullptr;
    uint16_t *cid_hot = nullptr;
    uint32_t *cid_cold = nullptr;
    uint32_t *ptr_all = nullptr, *rowidx_all = nullptr, *tile_row_all = nullptr;
    uint32_t *hot_wg_off = nullptr;
    ColdGroup *groups = nullptr;
    uint32_t ngroups = 0, hot_wgs = 0, cold_blocks = 0, max_tiles = 0, short_first_block = 0;
    uint64_t total_tiles = 0;
    std::vector<BandPiece> host_pieces;            // carry / out filled per scratch
    std::vector<uint64_t> carry_off;
    std::unordered_map<void *, BandScratch> scratch;
    uint64_t bytes = 0;                            // HBM held by the plan (without scratch)
};

void band_free(BandPlan *bp) {
    if (!bp) return;
    auto drop = [](void *p) {
        if (p) (void)hipFree(p);
    };
    drop(bp->perm);
    drop(bp->long_rows);
    drop(bp->vals_hot);
    drop(bp->vals_cold);
    drop(bp->cid_hot);
    drop(bp->cid_cold);
    drop(bp->ptr_all);
    drop(bp->rowidx_all);
    drop(bp->tile_row_all);
    drop(bp->hot_wg_off);
    drop(bp->groups);
    for (auto &kv : bp->scratch) {
        drop(kv.second.partial);
        drop(kv.second.carry);
        drop(kv.second.xp);
        drop(kv.second.pieces);
    }
    delete bp;
}

namespace {

struct PlanGuard {
    BandPlan *p;
    ~PlanGuard() { band_free(p); }
};

template <typename IDX, typename PTR>
int32_t band_build_t(sprs_hip_csmat *a, hipStream_t stream, BandPlan **out) {
    const Options &o = options();
    const uint64_t rows = a->rows, cols = a->cols, nnz = a->nnz;
    const PTR *ip = (const PTR *)a->indptr;
    const IDX *ix = (const IDX *)a->indices;
    *out = nullptr;
    if (rows >= 0xFFFFFFFFull || cols >= 0xFFFFFFFFull || !nnz) return SPRS_HIP_OK;
    const uint64_t split = (uint64_t)o.spmv_xcs_split;

    // ---- row classes --------------------------------------------------------------------
    TmpBuf short_flag, short_len, long_flag, short_pos, short_ptr, long_pos;
    SPRS_TRY_HIP(short_flag.alloc(rows * 8));
    SPRS_TRY_HIP(short_len.alloc(rows * 8));
    SPRS_TRY_HIP(long_flag.alloc(rows * 8));
    SPRS_TRY_HIP(short_pos.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(short_ptr.alloc((rows + 1) * 8));
    SPRS_TRY_HIP(long_pos.alloc((rows + 1) * 8));
    hipLaunchKernelGGL(bp_classify_kernel<PTR>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, ip, rows, split,
                       short_flag.u64(), short_len.u64(), long_flag.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(short_flag.u64(), short_pos.u64(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(short_len.u64(), short_ptr.u64(), rows, stream));
    SPRS_TRY(exclusive_scan_u64(long_flag.u64(), long_pos.u64(), rows, stream));
    uint64_t n_short_rows = 0, nnz_short = 0, n_long = 0;
    SPRS_TRY_HIP(hipMemcpy(&n_short_rows, short_pos.u64() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&nnz_short, short_ptr.u64() + rows, 8, hipMemcpyDeviceToHost));
    SPRS_TRY_HIP(hipMemcpy(&n_long, long_pos.u64() + rows, 8, hipMemcpyDeviceToHost));
    if (!n_long) return SPRS_HIP_OK;
    // auto mode: only when the long rows carry most of the entries (as the XCD-sliced plan)
    if (o.spmv_band == 0 && (nnz - nnz_short) * 2 < nnz) return SPRS_HIP_OK;
    if (nnz - nnz_short >= 0xFFFFFFFFull || nnz_short >= 0xFFFFFFFFull) return SPRS_HIP_OK;   // 32-bit piece offsets

    BandPlan *bp = new BandPlan();
    PlanGuard guard{bp};
    bp->cols = cols;
    bp->cols_pad = (cols + CB - 1) / CB * CB + CB;
    bp->n_long = (uint32_t)n_long;
    bp->n_short_rows = (uint32_t)n_short_rows;
    bp->G = (uint32_t)(o.spmv_band_group > 0 ? o.spmv_band_group : 4);
    uint64_t nh = o.spmv_band_hot > 0 ? (uint64_t)o.spmv_band_hot : 24;
    if (nh > (cols + CB - 1) / CB) nh = (cols + CB - 1) / CB;
    if (nh > (uint64_t)MAX_HOT) nh = MAX_HOT;
    uint64_t phases = o.spmv_band_phases > 0 ? (uint64_t)o.spmv_band_phases : 1;
    if (phases > (uint64_t)MAX_PHASES) phases = MAX_PHASES;
    bp->nh = (uint32_t)nh;
    bp->phases = (uint32_t)phases;
    const uint32_t NP = (uint32_t)(nh + 8 * phases);
    bp->npieces = NP;
    SliceMap map;
    map.nh = (uint32_t)nh;
    map.phases = (uint32_t)phases;
    map.hot_labels = nh * CB;
    const uint64_t cold_labels = cols > map.hot_labels ? cols - map.hot_labels : 0;
    map.phase_width = (cold_labels + phases - 1) / phases;
    if (!map.phase_width) map.phase_width = 1;

    // ---- labels ---------------------------------------------------------------------------
    SPRS_TRY(build_column_labels<IDX>(ix, nnz, cols, stream, &bp->perm));

    // ---- short piece + list of long rows ------------------------------------------------------
    // cold arrays: [short piece | cold pieces], every piece starting at a multiple of 4 entries
    SPRS_TRY_HIP(hipMalloc((void **)&bp->long_rows, n_long * 4));

    // ---- long rows: count per piece, scans, placement --------------------------------------------
    const uint64_t flat = (uint64_t)NP * n_long;
    TmpBuf cnt, nz, pos, pair, starts_d, pb_d;
    SPRS_TRY_HIP(cnt.alloc(flat * 8));
    SPRS_TRY_HIP(nz.alloc(flat * 8));
    SPRS_TRY_HIP(pos.alloc((flat + 1) * 8));
    SPRS_TRY_HIP(pair.alloc((flat + 1) * 8));
    // long_rows is needed by the count kernel: fill it (and the short piece) first.  The short piece's arrays are
    // allocated below once the cold sizes are known, so the fill runs in two steps: rows first.
    // (bp_fill_short_kernel writes both; its s_* targets are allocated right here with the short sizes.)
    TmpBuf s_rowidx_t, s_ptr_t, s_cid_t, s_val_t;
    SPRS_TRY_HIP(s_rowidx_t.alloc((n_short_rows + 1) * 4));
    SPRS_TRY_HIP(s_ptr_t.alloc((n_short_rows + 1) * 4));
    SPRS_TRY_HIP(s_cid_t.alloc((nnz_short + 4) * 4));
    SPRS_TRY_HIP(s_val_t.alloc((nnz_short + 2) * 8));
    hipLaunchKernelGGL((bp_fill_short_kernel<IDX, PTR>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream, ip, ix,
                       a->data, rows, short_pos.u64(), short_ptr.u64(), long_pos.u64(), bp->perm, (uint32_t *)s_rowidx_t.p,
                       (uint32_t *)s_ptr_t.p, (uint32_t *)s_cid_t.p, (double *)s_val_t.p, bp->long_rows, split);
    SPRS_TRY_HIP(hipGetLastError());
    uint64_t wblocks = (n_long + 3) / 4;
    if (wblocks > 256 * 64) wblocks = 256 * 64;
    hipLaunchKernelGGL((bp_count_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(256), 0, stream, ip, ix, bp->long_rows,
                       n_long, bp->perm, map, NP, cnt.u64(), nz.u64());
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(exclusive_scan_u64(cnt.u64(), pos.u64(), flat, stream));
    SPRS_TRY(exclusive_scan_u64(nz.u64(), pair.u64(), flat, stream));
    SPRS_TRY_HIP(starts_d.alloc(2 * (NP + 1) * 8));
    hipLaunchKernelGGL(bp_starts_kernel, dim3((NP + 256) / 256), dim3(256), 0, stream, pos.u64(), pair.u64(), n_long, NP,
                       starts_d.u64());
    SPRS_TRY_HIP(hipGetLastError());
    std::vector<uint64_t> starts(2 * (NP + 1));
    SPRS_TRY_HIP(hipMemcpy(starts.data(), starts_d.p, starts.size() * 8, hipMemcpyDeviceToHost));

    std::vector<PieceBuild> pb(NP);
    bp->host_pieces.assign(NP + 1, BandPiece());
    bp->carry_off.assign(NP + 2, 0);
    uint64_t hot_tiles = 0, cold_ent = (nnz_short + 3) & ~3ull, ptr_off = 0, row_off = 0, tile_off = 0;
    std::vector<uint32_t> hot_wg_off(nh + 1, 0);
    uint32_t max_tiles = 0;
    for (uint32_t k = 0; k < NP; ++k) {
        PieceBuild &b = pb[k];
        b.start = starts[k];
        b.nnz = starts[k + 1] - starts[k];
        b.pair0 = starts[NP + 1 + k];
        const uint64_t nr = starts[NP + 1 + k + 1] - b.pair0;
        b.hot = k < nh ? 1u : 0u;
        b.x0 = k < nh ? k * CB : 0u;
        b.ptr_off = (uint32_t)ptr_off;
        b.row_off = (uint32_t)row_off;
        BandPiece &d = bp->host_pieces[k];
        d.nnz = b.nnz;
        d.nr = (uint32_t)nr;
        d.x0 = b.x0;
        d.to_y = 0;
        if (b.hot) {
            const uint64_t nblocks = (b.nnz + HT - 1) / HT;       // the arrays are padded to whole blocks of HT entries
            d.ntiles = (uint32_t)((b.nnz + WT - 1) / WT);         // wave tiles
            b.ent0 = hot_tiles * HT;
            hot_tiles += nblocks;
            hot_wg_off[k + 1] = hot_wg_off[k] + (uint32_t)((nblocks + bp->G - 1) / bp->G);
        } else {
            d.ntiles = (uint32_t)((b.nnz + CT - 1) / CT);
            b.ent0 = cold_ent;
            cold_ent = (cold_ent + b.nnz + 3) & ~3ull;
        }
        d.ent0 = b.ent0;
        bp->carry_off[k] = tile_off;     // carry slots and tile_row share the running tile count (+1 per piece for tile_row)
        ptr_off += nr + 1;
        row_off += nr;
        tile_off += d.ntiles + 1;
        if (d.ntiles > max_tiles) max_tiles = d.ntiles;
    }
    {   // the short piece: index NP
        BandPiece &d = bp->host_pieces[NP];
        d.nnz = nnz_short;
        d.nr = (uint32_t)n_short_rows;
        d.ntiles = (uint32_t)((nnz_short + CT - 1) / CT);
        d.ent0 = 0;
        d.x0 = 0;
        d.to_y = 1;
        bp->carry_off[NP] = tile_off;
        tile_off += d.ntiles + 1;
        bp->carry_off[NP + 1] = tile_off;
        if (d.ntiles > max_tiles) max_tiles = d.ntiles;
    }
    bp->max_tiles = max_tiles;
    bp->total_tiles = tile_off;
    bp->hot_wgs = hot_wg_off[nh];
    if (ptr_off + n_short_rows + 1 >= 0xFFFFFFFFull || tile_off >= 0x7FFFFFFFull) return SPRS_HIP_OK;

    // ---- arrays of the plan ----------------------------------------------------------------------
    const uint64_t hot_entries = hot_tiles * HT;
    SPRS_TRY_HIP(hipMalloc((void **)&bp->vals_hot, (hot_entries + 2) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->cid_hot, (hot_entries + 8) * 2));
    SPRS_TRY_HIP(hipMemsetAsync(bp->vals_hot, 0, (hot_entries + 2) * 8, stream));
    SPRS_TRY_HIP(hipMemsetAsync(bp->cid_hot, 0, (hot_entries + 8) * 2, stream));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->vals_cold, (cold_ent + CT) * 8));     // a last partial tile is read element-wise, never past nnz
    SPRS_TRY_HIP(hipMalloc((void **)&bp->cid_cold, (cold_ent + CT) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->ptr_all, (ptr_off + n_short_rows + 2) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->rowidx_all, (row_off + n_short_rows + 1) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->tile_row_all, (tile_off + 1) * 4));
    bp->bytes = (hot_entries + 2) * 10 + (cold_ent + CT) * 12 + (ptr_off + row_off + 2 * n_short_rows + tile_off) * 4 +
                cols * 4 + n_long * 4;
    // short piece: move the temporaries into place (device-to-device)
    if (nnz_short) {
        SPRS_TRY_HIP(hipMemcpyAsync(bp->vals_cold, s_val_t.p, nnz_short * 8, hipMemcpyDeviceToDevice, stream));
        SPRS_TRY_HIP(hipMemcpyAsync(bp->cid_cold, s_cid_t.p, nnz_short * 4, hipMemcpyDeviceToDevice, stream));
    }
    SPRS_TRY_HIP(hipMemcpyAsync(bp->ptr_all + ptr_off, s_ptr_t.p, (n_short_rows + 1) * 4, hipMemcpyDeviceToDevice, stream));
    if (n_short_rows)
        SPRS_TRY_HIP(hipMemcpyAsync(bp->rowidx_all + row_off, s_rowidx_t.p, n_short_rows * 4, hipMemcpyDeviceToDevice, stream));

    SPRS_TRY_HIP(pb_d.alloc(NP * sizeof(PieceBuild)));
    SPRS_TRY_HIP(hipMemcpyAsync(pb_d.p, pb.data(), NP * sizeof(PieceBuild), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL((bp_scatter_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(256), 0, stream, ip, ix, a->data,
                       bp->long_rows, n_long, bp->perm, map, NP, pos.u64(), (const PieceBuild *)pb_d.p, bp->vals_hot,
                       bp->cid_hot, bp->vals_cold, bp->cid_cold);
    SPRS_TRY_HIP(hipGetLastError());
    hipLaunchKernelGGL(bp_rows_kernel, dim3((unsigned)((flat + 255) / 256)), dim3(256), 0, stream, cnt.u64(), pos.u64(),
                       pair.u64(), n_long, NP, (const PieceBuild *)pb_d.p, bp->ptr_all, bp->rowidx_all);
    SPRS_TRY_HIP(hipGetLastError());

    // ---- device pointers of the pieces, tile -> first row tables ------------------------------------
    std::vector<TileRowJob> jobs(NP + 1);
    for (uint32_t k = 0; k <= NP; ++k) {
        BandPiece &d = bp->host_pieces[k];
        const uint64_t po = k < NP ? pb[k].ptr_off : ptr_off, ro = k < NP ? pb[k].row_off : row_off;
        d.ptr = bp->ptr_all + po;
        d.rowidx = bp->rowidx_all + ro;
        d.tile_row = bp->tile_row_all + bp->carry_off[k];
        jobs[k] = TileRowJob{d.ptr, bp->tile_row_all + bp->carry_off[k], d.nr, d.ntiles, k < nh ? (uint32_t)WT : (uint32_t)CT};
    }
    TmpBuf jobs_d;
    SPRS_TRY_HIP(jobs_d.alloc(jobs.size() * sizeof(TileRowJob)));
    SPRS_TRY_HIP(hipMemcpyAsync(jobs_d.p, jobs.data(), jobs.size() * sizeof(TileRowJob), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bp_tile_rows_kernel, dim3((max_tiles + 256) / 256, NP + 1), dim3(256), 0, stream,
                       (const TileRowJob *)jobs_d.p);
    SPRS_TRY_HIP(hipGetLastError());

    // ---- launch tables -------------------------------------------------------------------------------
    SPRS_TRY_HIP(hipMalloc((void **)&bp->hot_wg_off, (nh + 1) * 4));
    SPRS_TRY_HIP(hipMemcpyAsync(bp->hot_wg_off, hot_wg_off.data(), (nh + 1) * 4, hipMemcpyHostToDevice, stream));
    std::vector<ColdGroup> groups;
    uint32_t blocks = 0;
    for (uint32_t ph = 0; ph < phases; ++ph) {
        uint32_t mt = 0;
        for (uint32_t s = 0; s < 8; ++s) mt = std::max(mt, bp->host_pieces[nh + ph * 8 + s].ntiles);
        if (!mt) continue;
        groups.push_back(ColdGroup{blocks, (uint32_t)(nh + ph * 8), 8});
        blocks += mt * 8;
    }
    if (bp->host_pieces[NP].ntiles) {
        bp->short_first_block = blocks;
        groups.push_back(ColdGroup{blocks, NP, 1});
        blocks += bp->host_pieces[NP].ntiles;
    }
    bp->ngroups = (uint32_t)groups.size();
    bp->cold_blocks = blocks;
    if (bp->ngroups) {
        SPRS_TRY_HIP(hipMalloc((void **)&bp->groups, groups.size() * sizeof(ColdGroup)));
        SPRS_TRY_HIP(hipMemcpyAsync(bp->groups, groups.data(), groups.size() * sizeof(ColdGroup), hipMemcpyHostToDevice, stream));
    }
    SPRS_TRY_HIP(hipStreamSynchronize(stream));   // plan complete, temporaries may go
    if (getenv("SPRS_HIP_DEBUG")) {
        uint64_t hot_nnz = 0, cold_nnz = 0, hot_pairs = 0, cold_pairs = 0;
        for (uint32_t k = 0; k < NP; ++k) {
            (k < nh ? hot_nnz : cold_nnz) += bp->host_pieces[k].nnz;
            (k < nh ? hot_pairs : cold_pairs) += bp->host_pieces[k].nr;
        }
        fprintf(stderr, "[sprs_hip band] rows %llu nnz %llu | long rows %llu, short non-empty rows %llu with %llu entries | hot: %u slices, %llu entries, "
                        "%llu (row,slice) pairs | cold: %u pieces, %llu entries, %llu pairs | plan %.1f MB\n",
                (unsigned long long)rows, (unsigned long long)nnz, (unsigned long long)n_long, (unsigned long long)n_short_rows,
                (unsigned long long)nnz_short, (unsigned)nh, (unsigned long long)hot_nnz, (unsigned long long)hot_pairs,
                (unsigned)(8 * phases), (unsigned long long)cold_nnz, (unsigned long long)cold_pairs, bp->bytes / 1e6);
        fprintf(stderr, "[sprs_hip band] entries per hot slice:");
        for (uint32_t k = 0; k < nh; ++k) fprintf(stderr, " %llu", (unsigned long long)bp->host_pieces[k].nnz);
        fprintf(stderr, "\n[sprs_hip band] entries per cold piece:");
        for (uint32_t k = nh; k < NP; ++k) fprintf(stderr, " %llu", (unsigned long long)bp->host_pieces[k].nnz);
        fprintf(stderr, "\n");
    }
    guard.p = nullptr;
    *out = bp;
    return SPRS_HIP_OK;
}

int32_t band_scratch(BandPlan *bp, hipStream_t stream, BandScratch **out) {
    auto it = bp->scratch.find((void *)stream);
    if (it == bp->scratch.end()) {
        BandScratch sc;
        const uint64_t pbytes = (uint64_t)bp->npieces * bp->n_long * 8;
        SPRS_TRY_HIP(hipMalloc((void **)&sc.partial, pbytes ? pbytes : 8));
        // rows without entries in a piece are never written by it: their partials must read as zero
        SPRS_TRY_HIP(hipMemset(sc.partial, 0, pbytes ? pbytes : 8));
        SPRS_TRY_HIP(hipMalloc((void **)&sc.carry, (bp->total_tiles + 1) * 8));
        SPRS_TRY_HIP(hipMalloc((void **)&sc.xp, bp->cols_pad * 8));
        SPRS_TRY_HIP(hipMemset(sc.xp, 0, bp->cols_pad * 8));      // the padding behind the last column is read into LDS
        std::vector<BandPiece> pcs = bp->host_pieces;
        for (uint32_t k = 0; k <= bp->npieces; ++k) {
            pcs[k].carry = sc.carry + bp->carry_off[k];
            pcs[k].out = k < bp->npieces ? sc.partial + (uint64_t)k * bp->n_long : nullptr;
        }
        SPRS_TRY_HIP(hipMalloc((void **)&sc.pieces, pcs.size() * sizeof(BandPiece)));
        SPRS_TRY_HIP(hipMemcpy(sc.pieces, pcs.data(), pcs.size() * sizeof(BandPiece), hipMemcpyHostToDevice));
        it = bp->scratch.emplace((void *)stream, sc).first;
    }
    *out = &it->second;
    return SPRS_HIP_OK;
}

}  // namespace

int32_t band_build(sprs_hip_csmat *a, hipStream_t stream, BandPlan **out) {
    if (a->idx_bytes == 8 && a->iptr_bytes == 8) return band_build_t<uint64_t, uint64_t>(a, stream, out);
    if (a->idx_bytes == 4 && a->iptr_bytes == 8) return band_build_t<uint32_t, uint64_t>(a, stream, out);
    if (a->idx_bytes == 8 && a->iptr_bytes == 4) return band_build_t<uint64_t, uint32_t>(a, stream, out);
    return band_build_t<uint32_t, uint32_t>(a, stream, out);
}

int32_t band_prepare(BandPlan *bp, hipStream_t stream) {
    BandScratch *sc = nullptr;
    return band_scratch(bp, stream, &sc);
}

uint64_t band_plan_bytes(const BandPlan *bp) { return bp ? bp->bytes : 0; }

// One SpMV on a banded plan.  The caller holds the handle's lock while the scratch is looked up (band_prepare).
int32_t band_spmv(sprs_hip_csmat *a, BandPlan *bp, const double *x, double *y, bool acc, hipStream_t stream) {
    BandScratch *sc = nullptr;
    {
        std::lock_guard<std::mutex> lock(a->mu);
        SPRS_TRY(band_scratch(bp, stream, &sc));
    }
    hipLaunchKernelGGL(rl_permute_x_kernel, dim3((unsigned)((bp->cols + 255) / 256)), dim3(256), 0, stream, x, bp->perm,
                       bp->cols, sc->xp);
    SPRS_TRY_HIP(hipGetLastError());
    if (!acc) SPRS_TRY_HIP(hipMemsetAsync(y, 0, a->rows * sizeof(double), stream));   // empty rows; the others are overwritten
    if (bp->hot_wgs) {
        hipLaunchKernelGGL(band_hot_kernel, dim3(bp->hot_wgs), dim3(HNT), 0, stream, (const BandPiece *)sc->pieces,
                           (const uint32_t *)bp->hot_wg_off, bp->nh, bp->G, (const double *)bp->vals_hot,
                           (const uint16_t *)bp->cid_hot, (const double *)sc->xp);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (bp->cold_blocks) {
        // one launch for the cold pieces and the short rows; option spmv_band_split_launch: two launches (profiling)
        uint32_t cut = bp->cold_blocks;
        if (options().spmv_band_split_launch && bp->short_first_block && bp->short_first_block < bp->cold_blocks)
            cut = bp->short_first_block;
        for (uint32_t part = 0; part < 2; ++part) {
            const uint32_t b0 = part ? cut : 0u, nb = part ? bp->cold_blocks - cut : cut;
            if (!nb) continue;
            if (acc)
                hipLaunchKernelGGL(band_cold_kernel<true>, dim3(nb), dim3(CNT), 0, stream, (const BandPiece *)sc->pieces,
                                   (const ColdGroup *)bp->groups, bp->ngroups, (const double *)bp->vals_cold,
                                   (const uint32_t *)bp->cid_cold, (const double *)sc->xp, y, b0);
            else
                hipLaunchKernelGGL(band_cold_kernel<false>, dim3(nb), dim3(CNT), 0, stream, (const BandPiece *)sc->pieces,
                                   (const ColdGroup *)bp->groups, bp->ngroups, (const double *)bp->vals_cold,
                                   (const uint32_t *)bp->cid_cold, (const double *)sc->xp, y, b0);
            SPRS_TRY_HIP(hipGetLastError());
        }
    }
    if (bp->max_tiles > 1) {
        hipLaunchKernelGGL(band_carry_kernel, dim3((bp->max_tiles + 255) / 256, bp->npieces + 1), dim3(256), 0, stream,
                           (const BandPiece *)sc->pieces, bp->nh, y);
        SPRS_TRY_HIP(hipGetLastError());
    }
    const dim3 rg((bp->n_long + 255) / 256), rb(256);
    if (acc)
        hipLaunchKernelGGL(band_reduce_kernel<true>, rg, rb, 0, stream, (const double *)sc->partial,
                           (const uint32_t *)bp->long_rows, y, bp->n_long, bp->npieces);
    else
        hipLaunchKernelGGL(band_reduce_kernel<false>, rg, rb, 0, stream, (const double *)sc->partial,
                           (const uint32_t *)bp->long_rows, y, bp->n_long, bp->npieces);
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
