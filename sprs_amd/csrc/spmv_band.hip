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
//   through carries — the machinery of spmv.hip.  A piece writes one partial sum per (row, piece) pair
//   it has, COMPACTLY, in the order of its own row list (neighbouring lanes write neighbouring words: the
//   first version scattered them into dense per-piece slabs, one 8-byte store per cache line, and those
//   stores cost as much as the gathers they had replaced).  A last kernel adds the partials of a long row
//   in piece order, finding them through per-(64 rows, piece) presence masks and base offsets.
//   No float atomics: bit-reproducible run to run.
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
constexpr int CNT = 256;             // threads of a cold / short workgroup: 4 independent waves, one wave tile each
constexpr int MAX_HOT = 384;
constexpr int MAX_PHASES = 8;
constexpr int MAX_PIECES = MAX_HOT + 8 * MAX_PHASES + 1;

// One CSR piece of the plan, as the kernels see it.
struct BandPiece {
    const uint32_t *ptr;        // nr + 1 entry offsets, relative to the piece
    const uint32_t *rowidx;     // nr: long-row number (short piece: row of y) of compact row r
    const uint32_t *tile_row;   // ntiles + 1: first compact row starting at / after tile c
    double *carry;              // ntiles
    double *out;                // nr partial sums, one per compact row, in row-list order; the short piece writes y[rowidx[r]] instead
    uint64_t ent0;              // first entry of the piece in the value / column-id arrays of its class
    uint64_t nnz;
    uint32_t nr, ntiles;
    uint32_t x0;                // hot: first label of the slice
    uint32_t to_y;              // short piece
};

// What a kernel keeps of a piece: the same fields with the pointers typed as GLOBAL memory.  A pointer loaded from a struct
// in memory is a generic ("flat") pointer to the compiler, and a flat load counts on the LDS counter as well: the
// tile_row load issued with the next tile's stream then held up the first s_waitcnt lgkmcnt(0) of the current tile's LDS
// work for a whole memory round trip — the software pipeline of the hot kernel did not overlap anything (round 2, found
// in the ISA: 13 flat operations in band_hot_kernel, 19 in band_cold_kernel).
#ifdef SPRS_HIP_EMU
#define SPRS_GLOBAL_AS
#else
#define SPRS_GLOBAL_AS __attribute__((address_space(1)))
#endif
struct PieceView {
    const SPRS_GLOBAL_AS uint32_t *rowidx, *tile_row;
    SPRS_GLOBAL_AS double *carry, *out;
    uint64_t ent0, nnz;
    uint32_t nr, ntiles, x0, to_y;
    __device__ __forceinline__ PieceView(const BandPiece &p)
        : rowidx((const SPRS_GLOBAL_AS uint32_t *)p.rowidx), tile_row((const SPRS_GLOBAL_AS uint32_t *)p.tile_row),
          carry((SPRS_GLOBAL_AS double *)p.carry), out((SPRS_GLOBAL_AS double *)p.out), ent0(p.ent0), nnz(p.nnz), nr(p.nr),
          ntiles(p.ntiles), x0(p.x0), to_y(p.to_y) {}
};

struct ColdGroup {              // a run of blocks of the cold launch
    uint32_t first_block, first_piece, npieces;   // npieces 8: block b -> piece b % 8 (XCD b % 8), tile b / 8; 1: tile b
};

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
constexpr int WPASS = WT / (WAVE * 2);        // 16-byte value loads per lane and tile
constexpr int EPL = WT / WAVE;                // entries per lane
constexpr uint32_t ROW_START = 0x8000u;       // flag bit in a hot column id
static_assert(WPASS == 4 && EPL == 8 && HT % WT == 0, "hot kernel geometry");

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave complete in order; this keeps the COMPILER from moving them across the hand-over
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// NTH threads = NTH / 64 waves per workgroup; a workgroup advances NTH / 64 wave tiles ("a block") per iteration.
// 1024 threads: one workgroup per CU; 512: two, so that the x-tile load of one overlaps the streaming of the other.
// WIDE = false: a hot slice (16-bit local ids, every entry's x comes from the LDS tile, partial sums out).
// WIDE = true:  the short-rows piece run the same way (one "slice" whose x tile holds the 8192 most referenced
//               labels): 32-bit labels laid out as for band_cold_kernel; an entry whose label lies in the tile reads
//               LDS, the others gather from xp; sums go to y[rowidx[...]].  On R-MAT 10M about a third of the short
//               rows' entries hit the tile — a third fewer 128-byte line fills, which is what bounds the gathers
//               (141 G lines/s = 16 channels x 64 B/clk per XCD; profiles/r02f-j).
template <int NTH, bool WIDE, bool ACC>
__global__ __launch_bounds__(NTH) void band_hot_kernel(const BandPiece *__restrict__ pieces,
                                                       const uint32_t *__restrict__ wg_off, uint32_t nh, uint32_t G,
                                                       const double *__restrict__ vals, const void *__restrict__ cid_any,
                                                       const double *__restrict__ xp, double *__restrict__ y) {
    const uint16_t *cid = (const uint16_t *)cid_any;            // WIDE: const uint32_t *
    __shared__ __attribute__((aligned(16))) double xs[CB];
    constexpr int WPB = NTH / WAVE;               // wave tiles per block = waves per workgroup
    __shared__ __attribute__((aligned(16))) double stage_s[WPB][WT];   // sums of the rows starting in the wave's tile
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t k = 0;
    while (k + 1 < nh && blockIdx.x >= wg_off[k + 1]) ++k;       // block-uniform
    const PieceView d(pieces[k]);
    const uint32_t nwt = d.ntiles;                               // wave tiles of the slice
    const uint32_t b0 = (blockIdx.x - wg_off[k]) * G;            // first block of this workgroup
    // x tile of the slice -> LDS (xp is padded to a whole number of tiles)
    {
        const dbl2 *src = (const dbl2 *)(xp + d.x0);
        dbl2 v[CB / (2 * NTH)];
#pragma unroll
        for (int q = 0; q < CB / (2 * NTH); ++q) v[q] = src[q * NTH + tid];
#pragma unroll
        for (int q = 0; q < CB / (2 * NTH); ++q) *(dbl2 *)&xs[2 * (q * NTH + tid)] = v[q];
    }
    double *stage = stage_s[wave];

    // pipeline registers: the next tile of this wave (stream in flight), tile_row of the one after
    dbl2 av[WPASS];
    u32x4 cw, cw2;
    uint32_t R0n = 0, R0nn = 0;
    auto tile_of = [&](uint32_t it) { return (b0 + it) * (uint32_t)WPB + wave; };
    auto request_row = [&](uint32_t w) { return d.tile_row[w < nwt ? w : nwt - 1]; };   // past the slice: harmless reload
    auto request_tile = [&](uint32_t w) {
        const uint64_t g = d.ent0 + (uint64_t)(w < nwt ? w : nwt - 1) * WT;
#pragma unroll
        for (int p = 0; p < WPASS; ++p) av[p] = __builtin_nontemporal_load((const dbl2 *)(vals + g + p * (WAVE * 2) + lane * 2));
        if constexpr (WIDE) {
            const uint32_t *c32 = (const uint32_t *)cid_any;
            cw = __builtin_nontemporal_load((const u32x4 *)(c32 + g + lane * 4));
            cw2 = __builtin_nontemporal_load((const u32x4 *)(c32 + g + WAVE * 4 + lane * 4));
        } else {
            cw = __builtin_nontemporal_load((const u32x4 *)(cid + g + lane * EPL));
        }
    };
    if (tile_of(0) < nwt) {                                      // (nwt > 0 for every launched workgroup)
        R0n = request_row(tile_of(0));
        request_tile(tile_of(0));
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
        if constexpr (WIDE) {
            double xv[EPL];
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                const uint32_t c = q < 4 ? cw[q % 4] : cw2[q % 4];
                const uint32_t lab = c & 0x7FFFFFFFu;
                xv[q] = lab < (uint32_t)CB ? xs[lab] : xp[lab];
                fb |= (c >> 31) << q;
            }
#pragma unroll
            for (int q = 0; q < EPL; ++q) pr[q] = lane * EPL + q < cnt ? av[q / 2][q % 2] * xv[q] : 0.0;
        } else {
#pragma unroll
            for (int p = 0; p < WPASS; ++p) {
                const uint32_t c2 = cw[p];
                const uint32_t i0 = lane * EPL + 2 * p;
                pr[2 * p] = i0 < cnt ? av[p][0] * xs[c2 & (CB - 1)] : 0.0;
                pr[2 * p + 1] = i0 + 1 < cnt ? av[p][1] * xs[(c2 >> 16) & (CB - 1)] : 0.0;
                fb |= ((c2 >> 15) & 1u) << (2 * p);
                fb |= ((c2 >> 31) & 1u) << (2 * p + 1);
            }
        }
        const uint32_t R0 = R0n;                                 // first compact row starting in this tile
        // ---- requests for the next tile of this wave ----------------------------------------------------
        R0n = R0nn;
        if (it + 1 < G && tile_of(it + 1) < nwt) {
            request_tile(tile_of(it + 1));
            R0nn = request_row(tile_of(it + 2));
        }
        // ---- rows starting in lower lanes: the ordinal of this lane's first row inside the tile ----------
        uint32_t prefix = 0, nf = 0;
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const unsigned long long m = __ballot((fb >> q) & 1u);
            prefix += (uint32_t)__popcll(m & below);
            nf += (uint32_t)__popcll(m);
        }
        // ---- serial fold of the lane's entries -------------------------------------------------------------
        double run = 0.0, head = 0.0;
        uint32_t seen = 0;                                       // rows started in this lane so far
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            if ((fb >> q) & 1u) {
                if (seen == 0) head = run;                       // the run that was open when the lane began ends here
                else stage[prefix + seen - 1] = run;             // a row that lies inside the lane
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
            else stage[prefix - 1] = v;
        }
        if (lane == WAVE - 1) {                                  // the run still open at the end of the tile
            if (nf == 0) d.carry[w] = S;                         // no row starts in the tile: all of it is head
            else stage[nf - 1] = S;                              // partial sum of the last row starting here
        }
        wave_lds_fence();                                        // stage[0 .. nf) complete
        if constexpr (WIDE) {
            for (uint32_t j = lane; j < nf; j += WAVE) {
                const uint32_t r = d.rowidx[R0 + j];
                if constexpr (ACC) y[r] = y[r] + stage[j];       // every compact row has entries: empty rows are never touched (prod.rs:120-126)
                else y[r] = stage[j];
            }
        } else {
            for (uint32_t j = lane; j < nf; j += WAVE) d.out[R0 + j] = stage[j];   // coalesced: compact rows R0 .. R0 + nf - 1
        }
        wave_lds_fence();                                        // read before the next tile overwrites it
    }
}

// ---------------------------------------------------------------------------------------------
// cold pieces and the short rows: x gathered through L1 / L2 from xp, 32-bit labels.
//
// Same wave-tile scheme as the hot kernel (512 entries per wave, 8 consecutive entries per lane, row sums
// in registers, row starts flagged in the ids: bit 31), without an x tile: four independent waves per
// workgroup, 4 KiB of LDS each (output staging), so that six workgroups share a CU and thousands of gathers
// are in flight per CU.  (First version: 4096-entry tiles with the products staged in LDS and workgroup
// barriers, three workgroups per CU: 135 G gathers/s whatever the piece — with ~190 gathers in flight per CU
// at a few hundred ns each that is a latency bound, not a cache-throughput one; profiles/r02c-e.)
// Layout of wave tile w: values as in the hot slices (entry 8 l + 2 p + e at vals[512 w + 128 p + 2 l + e]);
// labels likewise transposed for two coalesced 16-byte loads: entry 8 l + 4 p + e at cid[512 w + 256 p + 4 l + e].
// Pieces start at multiples of 512 entries and are padded with zeros.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ROW_START32 = 0x80000000u;
// Cold ids in "natural" plans (option spmv_band_natural = 1, an experiment kept for A/B): an entry whose column is NOT one
// of the hot labels keeps its ORIGINAL column and carries this flag — the cold kernel then gathers it from the caller's x,
// and the per-SpMV permutation shrinks to the hot labels (a 12 us gather instead of a 10 M-element scatter).  Measured
// SLOWER on R-MAT 10M, 1.32 against 1.12 ms per SpMV (profiles/r03i): in the natural order a cold x line mixes columns of
// very different popularity, the cold kernel's gathers miss more (681 against 535 us) and take bandwidth from the hot
// kernel beside them (1143 against 975 us) — the labelling pays for the scatter several times over.
constexpr uint32_t NATURAL_ID = 0x40000000u;

// how the cold kernel reads x: 0 plain, 1 non-temporal, 2 device-scope (served by L2, no L1 allocation)
template <int POLICY>
__device__ __forceinline__ double gather_x(const double *__restrict__ xp, uint32_t i) {
    if constexpr (POLICY == 1) return __builtin_nontemporal_load(xp + i);
    else if constexpr (POLICY == 2) return __hip_atomic_load(xp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return xp[i];
}

template <bool ACC, int POLICY>
__global__ __launch_bounds__(CNT) void band_cold_kernel(const BandPiece *__restrict__ pieces,
                                                        const ColdGroup *__restrict__ groups, uint32_t ngroups,
                                                        const double *__restrict__ vals, const uint32_t *__restrict__ cid,
                                                        const double *__restrict__ xp, const double *__restrict__ x,
                                                        double *__restrict__ y, uint32_t block0) {
    constexpr int WPB = CNT / WAVE;
    __shared__ __attribute__((aligned(16))) double stage_s[WPB][WT];
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t bid = blockIdx.x + block0;
    uint32_t g = 0;
    while (g + 1 < ngroups && bid >= groups[g + 1].first_block) ++g;     // block-uniform
    const ColdGroup cg = groups[g];
    const uint32_t lb = bid - cg.first_block;
    const uint32_t pi = cg.first_piece + (cg.npieces == 1 ? 0u : (lb & 7u));
    const uint32_t w = (cg.npieces == 1 ? lb : (lb >> 3)) * WPB + wave;  // wave tile of the piece
    const PieceView d(pieces[pi]);
    if (w >= d.ntiles) return;                                           // wave-uniform; no workgroup barrier below
    double *stage = stage_s[wave];
    const uint64_t base = (uint64_t)w * WT;
    const uint32_t cnt = d.nnz - base < (uint64_t)WT ? (uint32_t)(d.nnz - base) : (uint32_t)WT;
    const uint64_t gpos = d.ent0 + base;
    dbl2 av[WPASS];
    u32x4 lw[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) lw[p] = __builtin_nontemporal_load((const u32x4 *)(cid + gpos + p * (WAVE * 4) + lane * 4));
#pragma unroll
    for (int p = 0; p < WPASS; ++p) av[p] = __builtin_nontemporal_load((const dbl2 *)(vals + gpos + p * (WAVE * 2) + lane * 2));
    const uint32_t R0 = d.tile_row[w];
    double xv[EPL];
    uint32_t fb = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const uint32_t c = lw[q / 4][q % 4];
        xv[q] = gather_x<POLICY>((c & NATURAL_ID) ? x : xp, c & ~(ROW_START32 | NATURAL_ID));   // padding: label 0, value 0, never summed into a row
        fb |= (c >> 31) << q;
    }
    double pr[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) pr[q] = lane * EPL + q < cnt ? av[q / 2][q % 2] * xv[q] : 0.0;
    // ---- rows starting in lower lanes ------------------------------------------------------------------
    uint32_t prefix = 0, nf = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const unsigned long long m = __ballot((fb >> q) & 1u);
        prefix += (uint32_t)__popcll(m & below);
        nf += (uint32_t)__popcll(m);
    }
    // ---- serial fold, segmented scan over the lanes (see band_hot_kernel) -----------------------------------
    double run = 0.0, head = 0.0;
    uint32_t seen = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        if ((fb >> q) & 1u) {
            if (seen == 0) head = run;
            else stage[prefix + seen - 1] = run;
            run = 0.0;
            ++seen;
        }
        run += pr[q];
    }
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
    double before = __shfl_up(S, 1, WAVE);
    if (lane == 0) before = 0.0;
    if (seen) {
        const double v = before + head;
        if (prefix == 0) d.carry[w] = v;
        else stage[prefix - 1] = v;
    }
    if (lane == WAVE - 1) {
        if (nf == 0) d.carry[w] = S;
        else stage[nf - 1] = S;
    }
    wave_lds_fence();                                                    // stage[0 .. nf) complete
    if (d.to_y) {
        for (uint32_t j = lane; j < nf; j += WAVE) {
            const uint32_t r = d.rowidx[R0 + j];
            if constexpr (ACC) y[r] = y[r] + stage[j];                   // every compact row has entries: empty rows are never touched (prod.rs:120-126)
            else y[r] = stage[j];
        }
    } else {
        for (uint32_t j = lane; j < nf; j += WAVE) d.out[R0 + j] = stage[j];
    }
}

// A row that spans several tiles of a piece gets the heads (carries) of the later tiles added, in tile order.
// Which rows those are is fixed by the plan: one SPILL record per such row, found once when the plan is built
// (scanning every tile of every piece in every SpMV took 33 us; profiles/r02c).
struct Spill {
    uint64_t out;               // index into the partial sums, or into y for the short piece
    uint32_t first, n;          // carry slots first .. first + n - 1
    uint32_t to_y, pad;
};

__global__ __launch_bounds__(256) void band_carry_kernel(const Spill *__restrict__ spills, uint32_t nspills,
                                                         const double *__restrict__ carry, double *__restrict__ partial,
                                                         double *__restrict__ y) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const bool valid = r < nspills;
    Spill sp = Spill{0, 0, 0, 0, 0};
    if (valid) sp = spills[r];
    // a short chain (nearly all: a row crossing one tile boundary) is added by its own thread ...
    const bool small = sp.n <= 8;
    if (valid && small) {
        double acc = 0.0;
        for (uint32_t i = 0; i < sp.n; ++i) acc += carry[sp.first + i];
        double *out = sp.to_y ? y : partial;
        out[sp.out] += acc;
    }
    // ... a long one (a hub row: hundreds of tiles) by the whole wave, lanes striding over it (fixed order)
    unsigned long long m = __ballot(valid && !small);
    while (m) {                                                  // wave-uniform
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t first = (uint32_t)__shfl((int)sp.first, b, WAVE), n = (uint32_t)__shfl((int)sp.n, b, WAVE);
        double acc = 0.0;
        for (uint32_t i = lane; i < n; i += WAVE) acc += carry[first + i];
        acc = wave_sum(acc);                                     // complete in lane 0
        const double tot = __shfl(acc, 0, WAVE);
        if (lane == (uint32_t)b) {
            double *out = sp.to_y ? y : partial;
            out[sp.out] += tot;
        }
    }
}

__global__ __launch_bounds__(256) void band_permute_kernel(const double *__restrict__ x, const uint32_t *__restrict__ perm,
                                                           uint64_t cols, double *__restrict__ xp, double *__restrict__ y_zero,
                                                           uint64_t rows, uint32_t first_label) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < cols) {
        const uint32_t l = perm[j];
        if (l >= first_label) xp[l] = x[j];     // labels below first_label were gathered by band_gather_hot_kernel
    }
    if (y_zero && j < rows) y_zero[j] = 0.0;
}

// The hot kernel only reads the labels of the hot slices: those are gathered first, through the inverse of the labelling
// (a few MB), so that the hot kernel starts a few us into the SpMV while the scatter of the other 90 % of x (and the clearing
// of y) runs beside it on the second stream, in front of the cold launch that needs them.
__global__ __launch_bounds__(256) void band_gather_hot_kernel(const double *__restrict__ x, const uint32_t *__restrict__ inv_hot,
                                                              uint32_t hot_labels, double *__restrict__ xp) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= hot_labels) return;
    const uint32_t j = inv_hot[l];
    xp[l] = j != 0xFFFFFFFFu ? x[j] : 0.0;
}

__global__ __launch_bounds__(256) void bp_inverse_hot_kernel(const uint32_t *__restrict__ perm, uint64_t cols, uint32_t hot_labels,
                                                             uint32_t *__restrict__ inv_hot) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    const uint32_t l = perm[j];
    if (l < hot_labels) inv_hot[l] = (uint32_t)j;
}

// y[long_rows[j]] (+)= sum of the row's partials.  One WORKGROUP per 64 consecutive long rows; its eight waves
// take the pieces k = q, q + 8, q + 16, ... (q = wave) and the eight sums are added in wave order: a fixed
// summation order, deterministic.  For piece k, wmask tells which of the 64 rows have a partial there and
// wbase where the first of them sits; the present rows' partials follow each other in memory, so the loads
// of a wave are contiguous.  Measured on R-MAT 10M (50 M partials, 128 hot slices): 140 us whatever the shape of the
// loop — one wave per row block, four or eight waves, 16 .. 36 loads in flight, eight row blocks per workgroup with
// the next tables prefetched (156 us) — because it is HBM-bound: 5.7 M L2 misses = 0.73 GB for 0.40 GB of
// partials, at the same ~41 G misses/s the streaming kernels reach (profiles/r02i .. r02o).
constexpr int RNW = 4;       // waves per row block in the reduction
// Table slot of piece k for row block wb: the pieces of wave q (k = q, q + RNW, ...) are stored next to each other.
__host__ __device__ __forceinline__ uint64_t reduce_slot(uint64_t wb, uint32_t k, uint32_t npieces) {
    const uint32_t nqmax = (npieces + RNW - 1) / RNW;
    return wb * ((uint64_t)nqmax * RNW) + (uint64_t)(k % RNW) * nqmax + k / RNW;
}

template <bool ACC>
__global__ __launch_bounds__(RNW * WAVE) void band_reduce_kernel(const double *__restrict__ partial,
                                                          const unsigned long long *__restrict__ wmask,
                                                          const uint32_t *__restrict__ wbase,
                                                          const uint32_t *__restrict__ long_rows, double *__restrict__ y,
                                                          uint32_t n_long, uint32_t npieces, uint32_t amask) {
    constexpr int NW = RNW;
    __shared__ double red[NW][WAVE];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));   // wave-uniform, kept in a scalar register
    // Block b runs on XCD b % 8 (observed; only speed depends on it): give every XCD a CONTIGUOUS range of row blocks
    // (neighbouring row blocks read neighbouring partials of every piece, often the same 128-byte line).
    const uint64_t nb = gridDim.x, qq = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const uint64_t wb = xcd * qq + (xcd < rem ? xcd : rem) + jj;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t nqmax = (npieces + NW - 1) / NW;
    const uint32_t nq = npieces > q ? (npieces - q + NW - 1) / NW : 0u;     // pieces of this wave: k = q + NW i, i < nq
    const uint64_t j = wb * WAVE + lane;
    const uint32_t r = long_rows[j < n_long ? j : n_long - 1];  // (requested early, unconditionally: needed only at the very end)
    // The wave's table rows are contiguous: ONE coalesced load puts the rows of 64 pieces into the lanes (lane l: piece
    // i0 + l), scalar broadcasts hand them out.  (Indexing the tables with the wave-uniform piece number compiles to one
    // scalar load plus a wait PER PIECE: a chain of 72 round trips.)
    const unsigned long long *mrow = wmask + wb * ((uint64_t)nqmax * NW) + (uint64_t)q * nqmax;
    const uint32_t *brow = wbase + wb * ((uint64_t)nqmax * NW) + (uint64_t)q * nqmax;
    double s = 0.0;
    constexpr int U = 36;                                       // partials in flight per lane: one chunk up to 144 pieces
    for (uint32_t i0 = 0; i0 < nq; i0 += WAVE) {
        const bool in = i0 + lane < nq;
        const unsigned long long mk = in ? mrow[i0 + lane] : 0ull;
        const uint32_t bs = in ? brow[i0 + lane] : 0u;
        const uint32_t nk = nq - i0 < (uint32_t)WAVE ? nq - i0 : (uint32_t)WAVE;
        for (uint32_t kk = 0; kk < nk; kk += U) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = (int)(kk + u < nk ? kk + u : nk - 1);                          // wave-uniform
                const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mk >> 32), src) << 32) |
                                             (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, src);
                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bs, src);
                const bool have = kk + u < nk && ((m >> lane) & 1ull);
                v[u] = have ? partial[(b + (uint32_t)__popcll(m & below)) & amask] : 0.0;   // amask: all ones (timing experiments: see spmv_xmask)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) s += v[u];              // ascending pieces (absent ones add +0.0)
        }
    }
    red[q][lane] = s;
    __syncthreads();
    if (q != 0) return;
    if (j >= n_long) return;
    double tot = red[0][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += red[w][lane];
    if constexpr (ACC) y[r] = y[r] + tot;
    else y[r] = tot;
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
                                     uint32_t *__restrict__ long_rows, uint64_t split, uint32_t natural_from) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    if (r == rows) {
        if (!s_cid) s_ptr[short_pos[rows]] = (uint32_t)short_ptr[rows];
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
    if (!s_cid) {                 // first call: the row lists
        s_rowidx[i] = (uint32_t)r;
        s_ptr[i] = (uint32_t)d;
        return;
    }
    for (uint64_t p = s; p < e; ++p, ++d) {       // second call: the entries, in the wave-tile layout of band_cold_kernel
        const uint64_t tile = d / WT;
        const uint32_t t = (uint32_t)(d % WT);
        const uint32_t ll = t / EPL, q = t % EPL;
        const uint32_t label = perm[indices[p]];
        const uint32_t id = label >= natural_from ? (uint32_t)indices[p] | NATURAL_ID : label;   // natural_from = 0xFFFFFFFF: labels only
        s_cid[tile * WT + (q / 4) * (WAVE * 4) + ll * 4 + (q & 3u)] = id | (p == s ? ROW_START32 : 0u);
        s_val[tile * WT + (q / 2) * (WAVE * 2) + ll * 2 + (q & 1u)] = data[p];
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
                                                         double *__restrict__ vals_cold, uint32_t *__restrict__ cid_cold,
                                                         uint32_t natural_from) {
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
            const uint32_t natural = valid ? (uint32_t)indices[p] | NATURAL_ID : 0u;
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
                    const uint64_t tile = e_rel / WT;
                    const uint32_t i = (uint32_t)(e_rel % WT);
                    const uint32_t ll = i / EPL, q = i % EPL;
                    vals_cold[b.ent0 + tile * WT + (q / 2) * (WAVE * 2) + ll * 2 + (q & 1u)] = v;
                    cid_cold[b.ent0 + tile * WT + (q / 4) * (WAVE * 4) + ll * 4 + (q & 3u)] =
                        (label >= natural_from ? natural : label) | (before + rank == 0 ? ROW_START32 : 0u);
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

// presence mask and first partial of every (block of 64 long rows, piece)
__global__ __launch_bounds__(256) void bp_wave_tables_kernel(const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ pair,
                                                             uint64_t n_long, uint32_t npieces, uint64_t nwb,
                                                             const PieceBuild *__restrict__ pb,
                                                             unsigned long long *__restrict__ wmask, uint32_t *__restrict__ wbase) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint64_t wid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;     // = wb * npieces + k
    if (wid >= nwb * npieces) return;                                                  // wave-uniform
    const uint64_t wb = wid / npieces;
    const uint32_t k = (uint32_t)(wid - wb * npieces);
    const uint64_t j = wb * WAVE + lane;
    const bool have = j < n_long && cnt[(uint64_t)k * n_long + j] != 0;
    const unsigned long long m = __ballot(have);
    if (lane == 0) {
        const uint64_t slot = reduce_slot(wb, k, npieces);
        wmask[slot] = m;
        wbase[slot] = pb[k].row_off + (uint32_t)(pair[(uint64_t)k * n_long + wb * WAVE] - pb[k].pair0);
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

struct SpillJob {
    const uint32_t *ptr, *rowidx, *tile_row;
    uint64_t out0;               // first partial sum of the piece; 0 for the short piece
    uint32_t ntiles, T, carry0, to_y;
};

// tile c of a piece spills when its last starting row runs on into tile c + 1
__global__ void bp_spill_kernel(const SpillJob *__restrict__ jobs, Spill *__restrict__ spills, unsigned int *__restrict__ count) {
    const SpillJob jb = jobs[blockIdx.y];
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c + 1 >= jb.ntiles) return;
    const uint32_t R0 = jb.tile_row[c], R1 = jb.tile_row[c + 1];
    if (R1 == R0) return;                                              // no row starts in tile c
    if ((uint64_t)jb.ptr[R1] <= (uint64_t)(c + 1) * jb.T) return;      // its last row ends inside tile c
    uint32_t n = 0;
    for (uint32_t e = c + 1; e < jb.ntiles && jb.tile_row[e] == R1; ++e) ++n;
    const unsigned int slot = atomicAdd(count, 1u);
    spills[slot] = Spill{jb.to_y ? (uint64_t)jb.rowidx[R1 - 1] : jb.out0 + (R1 - 1), jb.carry0 + c + 1, n, jb.to_y, 0u};
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct BandScratch {        // per stream
    double *partial = nullptr, *carry = nullptr, *xp = nullptr;
    BandPiece *pieces = nullptr;
    // the gather-bound kernels (cold pieces, short rows: L2 -> L1 fills) run beside the HBM-bound hot kernel on a second stream
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

struct BandPlan {
    uint32_t nh = 0, phases = 1, npieces = 0;      // npieces = nh + 8 phases (the short piece comes after them)
    uint32_t n_long = 0, n_short_rows = 0, G = 4;
    uint32_t hot_threads = 1024, hot_block = 8192; // threads of a hot workgroup; entries it advances per iteration
    uint64_t cols = 0, cols_pad = 0;
    uint32_t *perm = nullptr, *long_rows = nullptr;
    bool natural = false;                          // cold entries carry their original column (NATURAL_ID) and read the caller's x
    uint32_t *inv_hot = nullptr;                   // column of each hot label (0xFFFFFFFF: label not in use)
    uint32_t hot_labels = 0;                       // nh * 8192, at most cols_pad
    double *vals_hot = nullptr, *vals_cold = nullptr;
    uint16_t *cid_hot = nullptr;
    uint32_t *cid_cold = nullptr;
    uint32_t *ptr_all = nullptr, *rowidx_all = nullptr, *tile_row_all = nullptr;
    uint32_t *hot_wg_off = nullptr;
    ColdGroup *groups = nullptr;
    unsigned long long *wmask = nullptr;           // per (64 long rows, piece): which rows have a partial
    uint32_t *wbase = nullptr;                     //                            and where the first one is
    uint64_t total_pairs = 0;
    void *spills = nullptr;                        // Spill records (device)
    uint32_t nspills = 0;
    uint32_t ngroups = 0, hot_wgs = 0, cold_blocks = 0, max_tiles = 0, short_first_block = 0;
    bool has_short_group = false;
    uint32_t short_wgs = 0, Gs = 4;                // workgroups of the tiled short-rows launch, blocks per workgroup
    uint32_t *short_wg_off = nullptr;              // {0, short_wgs}
    uint64_t total_tiles = 0;
    std::vector<BandPiece> host_pieces;            // carry / out filled per scratch
    std::vector<uint64_t> carry_off, pair_off;
    std::unordered_map<void *, BandScratch> scratch;
    uint64_t bytes = 0;                            // HBM held by the plan (without scratch)
};

void band_free(BandPlan *bp) {
    if (!bp) return;
    auto drop = [](void *p) {
        if (p) (void)hipFree(p);
    };
    drop(bp->perm);
    drop(bp->inv_hot);
    drop(bp->long_rows);
    drop(bp->vals_hot);
    drop(bp->vals_cold);
    drop(bp->cid_hot);
    drop(bp->cid_cold);
    drop(bp->ptr_all);
    drop(bp->rowidx_all);
    drop(bp->tile_row_all);
    drop(bp->hot_wg_off);
    drop(bp->short_wg_off);
    drop(bp->groups);
    drop(bp->spills);
    drop(bp->wmask);
    drop(bp->wbase);
    for (auto &kv : bp->scratch) {
        drop(kv.second.partial);
        drop(kv.second.carry);
        drop(kv.second.xp);
        drop(kv.second.pieces);
        if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
        if (kv.second.join) (void)hipEventDestroy(kv.second.join);
        if (kv.second.aux) (void)hipStreamDestroy(kv.second.aux);
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
    if (rows >= 0xFFFFFFFFull || cols >= 0x7FFFFFFFull || !nnz) return SPRS_HIP_OK;   // bit 31 of a label flags a row start
    // rows with at least this many entries are "long" (cut into pieces); 24 measured best with 128 hot slices (r02h, r02o)
    const uint64_t split = o.spmv_band_split > 0 ? (uint64_t)o.spmv_band_split : 24ull;

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
    bp->hot_threads = o.spmv_band_hot_threads == 512 ? 512u : 1024u;
    bp->hot_block = bp->hot_threads / WAVE * WT;
    bp->G = (uint32_t)(o.spmv_band_group > 0 ? o.spmv_band_group : (bp->hot_threads == 1024 ? 16 : 32));   // 16 blocks = 131 072 entries per x-tile load
    uint64_t nh = o.spmv_band_hot > 0 ? (uint64_t)o.spmv_band_hot : 128;   // measured on R-MAT 10M: 48 .. 192 within 3 % (profiles/r02g, r02h)
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
    bp->hot_labels = (uint32_t)(map.hot_labels < bp->cols_pad ? map.hot_labels : bp->cols_pad);
    SPRS_TRY_HIP(hipMalloc((void **)&bp->inv_hot, ((uint64_t)bp->hot_labels + 1) * 4));
    SPRS_TRY_HIP(hipMemsetAsync(bp->inv_hot, 0xFF, ((uint64_t)bp->hot_labels + 1) * 4, stream));
    hipLaunchKernelGGL(bp_inverse_hot_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, (const uint32_t *)bp->perm,
                       cols, bp->hot_labels, bp->inv_hot);
    SPRS_TRY_HIP(hipGetLastError());
    bp->bytes += ((uint64_t)bp->hot_labels + 1) * 4;
    // natural ids need bit 30 free, and the (opt-in) tiled short-rows launch reads labels only
    bp->natural = o.spmv_band_natural == 1 && cols < (uint64_t)NATURAL_ID && o.spmv_band_short != 1 && bp->hot_labels != 0;
    const uint32_t natural_from = bp->natural ? bp->hot_labels : 0xFFFFFFFFu;

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
    TmpBuf s_rowidx_t, s_ptr_t;
    SPRS_TRY_HIP(s_rowidx_t.alloc((n_short_rows + 1) * 4));
    SPRS_TRY_HIP(s_ptr_t.alloc((n_short_rows + 1) * 4));
    hipLaunchKernelGGL((bp_fill_short_kernel<IDX, PTR>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream, ip, ix,
                       a->data, rows, short_pos.u64(), short_ptr.u64(), long_pos.u64(), bp->perm, (uint32_t *)s_rowidx_t.p,
                       (uint32_t *)s_ptr_t.p, (uint32_t *)nullptr, (double *)nullptr, bp->long_rows, split, 0xFFFFFFFFu);
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
    bp->pair_off.assign(NP + 1, 0);
    uint64_t hot_tiles = 0, cold_ent = (nnz_short + WT - 1) / WT * WT, ptr_off = 0, row_off = 0, tile_off = 0;   // the short piece comes first
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
        bp->pair_off[k] = row_off;
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
            const uint64_t wg_blocks = (b.nnz + bp->hot_block - 1) / bp->hot_block;   // blocks of one workgroup iteration
            hot_wg_off[k + 1] = hot_wg_off[k] + (uint32_t)((wg_blocks + bp->G - 1) / bp->G);
        } else {
            d.ntiles = (uint32_t)((b.nnz + WT - 1) / WT);         // wave tiles; the piece is padded to whole tiles
            b.ent0 = cold_ent;
            cold_ent += (uint64_t)d.ntiles * WT;
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
        d.ntiles = (uint32_t)((nnz_short + WT - 1) / WT);
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
    SPRS_TRY_HIP(hipMalloc((void **)&bp->vals_cold, (cold_ent + WT) * 8));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->cid_cold, (cold_ent + WT) * 4));
    SPRS_TRY_HIP(hipMemsetAsync(bp->vals_cold, 0, (cold_ent + WT) * 8, stream));   // the padding of every piece reads as (label 0, value 0)
    SPRS_TRY_HIP(hipMemsetAsync(bp->cid_cold, 0, (cold_ent + WT) * 4, stream));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->ptr_all, (ptr_off + n_short_rows + 2) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->rowidx_all, (row_off + n_short_rows + 1) * 4));
    SPRS_TRY_HIP(hipMalloc((void **)&bp->tile_row_all, (tile_off + 1) * 4));
    bp->bytes = (hot_entries + 2) * 10 + (cold_ent + WT) * 12 + (ptr_off + row_off + 2 * n_short_rows + tile_off) * 4 +
                cols * 4 + n_long * 4;
    // short piece: its entries go straight into place (piece 0 of the cold arrays), its row lists are copied
    hipLaunchKernelGGL((bp_fill_short_kernel<IDX, PTR>), dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, stream, ip, ix,
                       a->data, rows, short_pos.u64(), short_ptr.u64(), long_pos.u64(), bp->perm, (uint32_t *)nullptr,
                       (uint32_t *)nullptr, bp->cid_cold, bp->vals_cold, bp->long_rows, split, natural_from);
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY_HIP(hipMemcpyAsync(bp->ptr_all + ptr_off, s_ptr_t.p, (n_short_rows + 1) * 4, hipMemcpyDeviceToDevice, stream));
    if (n_short_rows)
        SPRS_TRY_HIP(hipMemcpyAsync(bp->rowidx_all + row_off, s_rowidx_t.p, n_short_rows * 4, hipMemcpyDeviceToDevice, stream));

    SPRS_TRY_HIP(pb_d.alloc(NP * sizeof(PieceBuild)));
    SPRS_TRY_HIP(hipMemcpyAsync(pb_d.p, pb.data(), NP * sizeof(PieceBuild), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL((bp_scatter_kernel<IDX, PTR>), dim3((unsigned)wblocks), dim3(256), 0, stream, ip, ix, a->data,
                       bp->long_rows, n_long, bp->perm, map, NP, pos.u64(), (const PieceBuild *)pb_d.p, bp->vals_hot,
                       bp->cid_hot, bp->vals_cold, bp->cid_cold, natural_from);
    SPRS_TRY_HIP(hipGetLastError());
    hipLaunchKernelGGL(bp_rows_kernel, dim3((unsigned)((flat + 255) / 256)), dim3(256), 0, stream, cnt.u64(), pos.u64(),
                       pair.u64(), n_long, NP, (const PieceBuild *)pb_d.p, bp->ptr_all, bp->rowidx_all);
    SPRS_TRY_HIP(hipGetLastError());

    {   // ---- tables of the final reduction ---------------------------------------------------------------
        const uint64_t nwb = (n_long + WAVE - 1) / WAVE;
        bp->total_pairs = row_off;
        const uint64_t slots = nwb * (uint64_t)((NP + RNW - 1) / RNW) * RNW;     // (padded: reduce_slot)
        SPRS_TRY_HIP(hipMalloc((void **)&bp->wmask, slots * 8));
        SPRS_TRY_HIP(hipMalloc((void **)&bp->wbase, slots * 4));
        SPRS_TRY_HIP(hipMemsetAsync(bp->wmask, 0, slots * 8, stream));
        SPRS_TRY_HIP(hipMemsetAsync(bp->wbase, 0, slots * 4, stream));
        const uint64_t waves = nwb * NP;
        hipLaunchKernelGGL(bp_wave_tables_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, cnt.u64(), pair.u64(),
                           n_long, NP, nwb, (const PieceBuild *)pb_d.p, bp->wmask, bp->wbase);
        SPRS_TRY_HIP(hipGetLastError());
        bp->bytes += nwb * NP * 12;
    }
    // ---- device pointers of the pieces, tile -> first row tables ------------------------------------
    std::vector<TileRowJob> jobs(NP + 1);
    for (uint32_t k = 0; k <= NP; ++k) {
        BandPiece &d = bp->host_pieces[k];
        const uint64_t po = k < NP ? pb[k].ptr_off : ptr_off, ro = k < NP ? pb[k].row_off : row_off;
        d.ptr = bp->ptr_all + po;
        d.rowidx = bp->rowidx_all + ro;
        d.tile_row = bp->tile_row_all + bp->carry_off[k];
        jobs[k] = TileRowJob{d.ptr, bp->tile_row_all + bp->carry_off[k], d.nr, d.ntiles, (uint32_t)WT};
    }
    TmpBuf jobs_d;
    SPRS_TRY_HIP(jobs_d.alloc(jobs.size() * sizeof(TileRowJob)));
    SPRS_TRY_HIP(hipMemcpyAsync(jobs_d.p, jobs.data(), jobs.size() * sizeof(TileRowJob), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bp_tile_rows_kernel, dim3((max_tiles + 256) / 256, NP + 1), dim3(256), 0, stream,
                       (const TileRowJob *)jobs_d.p);
    SPRS_TRY_HIP(hipGetLastError());

    // ---- rows that span tiles: one record each ----------------------------------------------------------
    {
        std::vector<SpillJob> sj(NP + 1);
        for (uint32_t k = 0; k <= NP; ++k) {
            const BandPiece &d = bp->host_pieces[k];
            sj[k] = SpillJob{d.ptr, d.rowidx, d.tile_row, k < NP ? (uint64_t)pb[k].row_off : 0ull, d.ntiles,
                             (uint32_t)WT, (uint32_t)bp->carry_off[k], k == NP ? 1u : 0u};
        }
        TmpBuf sj_d, cnt_d;
        SPRS_TRY_HIP(sj_d.alloc(sj.size() * sizeof(SpillJob)));
        SPRS_TRY_HIP(cnt_d.alloc(4));
        SPRS_TRY_HIP(hipMemcpyAsync(sj_d.p, sj.data(), sj.size() * sizeof(SpillJob), hipMemcpyHostToDevice, stream));
        SPRS_TRY_HIP(hipMemsetAsync(cnt_d.p, 0, 4, stream));
        SPRS_TRY_HIP(hipMalloc(&bp->spills, (tile_off + 1) * sizeof(Spill)));     // at most one per tile
        hipLaunchKernelGGL(bp_spill_kernel, dim3((max_tiles + 255) / 256, NP + 1), dim3(256), 0, stream,
                           (const SpillJob *)sj_d.p, (Spill *)bp->spills, (unsigned int *)cnt_d.p);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY_HIP(hipMemcpy(&bp->nspills, cnt_d.p, 4, hipMemcpyDeviceToHost));
        // keep only what is used
        if (bp->nspills < tile_off / 4) {
            void *small = nullptr;
            SPRS_TRY_HIP(hipMalloc(&small, ((uint64_t)bp->nspills + 1) * sizeof(Spill)));
            SPRS_TRY_HIP(hipMemcpy(small, bp->spills, (uint64_t)bp->nspills * sizeof(Spill), hipMemcpyDeviceToDevice));
            (void)hipFree(bp->spills);
            bp->spills = small;
        }
        bp->bytes += ((uint64_t)bp->nspills + 1) * sizeof(Spill);
    }
    // ---- launch tables -------------------------------------------------------------------------------
    SPRS_TRY_HIP(hipMalloc((void **)&bp->hot_wg_off, (nh + 1) * 4));
    SPRS_TRY_HIP(hipMemcpyAsync(bp->hot_wg_off, hot_wg_off.data(), (nh + 1) * 4, hipMemcpyHostToDevice, stream));
    std::vector<ColdGroup> groups;
    uint32_t blocks = 0;
    for (uint32_t ph = 0; ph < phases; ++ph) {
        uint32_t mt = 0;
        for (uint32_t s = 0; s < 8; ++s) mt = std::max(mt, (bp->host_pieces[nh + ph * 8 + s].ntiles + CNT / WAVE - 1) / (CNT / WAVE));
        if (!mt) continue;
        groups.push_back(ColdGroup{blocks, (uint32_t)(nh + ph * 8), 8});
        blocks += mt * 8;
    }
    if (bp->host_pieces[NP].ntiles) {
        bp->short_first_block = blocks;
        bp->has_short_group = true;
        groups.push_back(ColdGroup{blocks, NP, 1});
        blocks += (bp->host_pieces[NP].ntiles + CNT / WAVE - 1) / (CNT / WAVE);
    }
    bp->ngroups = (uint32_t)groups.size();
    bp->cold_blocks = blocks;
    {   // tiled launch of the short rows: blocks of 16 wave tiles, G blocks per workgroup
        const uint64_t sblocks = (nnz_short + HT - 1) / HT;
        bp->Gs = (uint32_t)(o.spmv_band_short_group > 0 ? o.spmv_band_short_group : 4);
        bp->short_wgs = (uint32_t)((sblocks + bp->Gs - 1) / bp->Gs);
        const uint32_t tab[2] = {0u, bp->short_wgs};
        SPRS_TRY_HIP(hipMalloc((void **)&bp->short_wg_off, sizeof tab));
        SPRS_TRY_HIP(hipMemcpyAsync(bp->short_wg_off, tab, sizeof tab, hipMemcpyHostToDevice, stream));
        SPRS_TRY_HIP(hipStreamSynchronize(stream));      // tab lives on this stack frame
    }
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
        SPRS_TRY_HIP(hipMalloc((void **)&sc.partial, (bp->total_pairs + 1) * 8));   // every pair is written by every SpMV
        SPRS_TRY_HIP(hipMalloc((void **)&sc.carry, (bp->total_tiles + 1) * 8));
        SPRS_TRY_HIP(hipMalloc((void **)&sc.xp, bp->cols_pad * 8));
        SPRS_TRY_HIP(hipMemset(sc.xp, 0, bp->cols_pad * 8));      // the padding behind the last column is read into LDS
        std::vector<BandPiece> pcs = bp->host_pieces;
        for (uint32_t k = 0; k <= bp->npieces; ++k) {
            pcs[k].carry = sc.carry + bp->carry_off[k];
            pcs[k].out = k < bp->npieces ? sc.partial + bp->pair_off[k] : nullptr;
        }
        SPRS_TRY_HIP(hipMalloc((void **)&sc.pieces, pcs.size() * sizeof(BandPiece)));
        SPRS_TRY_HIP(hipMemcpy(sc.pieces, pcs.data(), pcs.size() * sizeof(BandPiece), hipMemcpyHostToDevice));
        SPRS_TRY_HIP(hipStreamCreateWithFlags(&sc.aux, hipStreamNonBlocking));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&sc.fork, hipEventDisableTiming));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&sc.join, hipEventDisableTiming));
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
        std::lock_guard<std::recursive_mutex> lock(a->mu);
        SPRS_TRY(band_scratch(bp, stream, &sc));
    }
    // x into the plan's labelling; the same launch clears y (empty rows; the others are overwritten) unless accumulating.
    // The gather-bound launch (cold pieces + short rows: L2 -> L1 line fills) runs on a second stream beside the
    // HBM-bound hot slices (option spmv_band_overlap, 2 = off).  The two kernels do run concurrently and mostly trade
    // time one for one (1145 vs 1153 us, profiles/r02h), but the gather workgroups fill the start-up and tail bubbles
    // of the one-workgroup-per-CU hot kernel: 1.12 vs 1.16 ms per SpMV over repeated A/B runs (profiles/r02j, r02l).
    // With the overlap the permutation is split as well (option spmv_band_split_permute, 2 = off): the hot labels are
    // gathered first (a few us), the hot kernel starts, and the scatter of the rest + the clearing of y go to the second
    // stream in front of the cold launch — 47 us less on the critical path.
    const bool overlap = options().spmv_band_overlap != 2 && bp->hot_wgs && bp->cold_blocks;
    const bool split_permute = bp->natural || (overlap && options().spmv_band_split_permute != 2 && bp->hot_labels);
    const uint64_t span = acc ? bp->cols : (bp->cols > a->rows ? bp->cols : a->rows);
    hipStream_t cstream = overlap ? sc->aux : stream;
    if (split_permute)
        hipLaunchKernelGGL(band_gather_hot_kernel, dim3((bp->hot_labels + 255) / 256), dim3(256), 0, stream, x,
                           (const uint32_t *)bp->inv_hot, bp->hot_labels, sc->xp);
    else
        hipLaunchKernelGGL(band_permute_kernel, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, stream, x,
                           (const uint32_t *)bp->perm, bp->cols, sc->xp, acc ? (double *)nullptr : y, a->rows, 0u);
    SPRS_TRY_HIP(hipGetLastError());
    if (overlap) {
        SPRS_TRY_HIP(hipEventRecord(sc->fork, stream));           // the hot labels of xp (or all of it, and the cleared y) are ready
        SPRS_TRY_HIP(hipStreamWaitEvent(sc->aux, sc->fork, 0));
    }
    if (bp->natural) {
        // nothing else to permute: the cold entries read x itself; y is cleared for the empty rows and the direct writers
        if (!acc) SPRS_TRY_HIP(hipMemsetAsync(y, 0, a->rows * sizeof(double), cstream));
    } else if (split_permute) {
        hipLaunchKernelGGL(band_permute_kernel, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, cstream, x,
                           (const uint32_t *)bp->perm, bp->cols, sc->xp, acc ? (double *)nullptr : y, a->rows, bp->hot_labels);
        SPRS_TRY_HIP(hipGetLastError());
    }
    // the short rows: with the hottest x entries in LDS (band_hot_kernel<.., true, ..>, default) or as one more gather piece
    const bool short_tiled = options().spmv_band_short == 1 && bp->short_wgs != 0;   // measured slower than the gather piece (profiles/r02k, r02l): opt-in
    if (bp->cold_blocks) {
        // one launch for the cold pieces (and the short rows when they are not tiled); option spmv_band_split_launch: two launches (profiling)
        uint32_t cut = bp->cold_blocks, end = bp->cold_blocks;
        if ((short_tiled || options().spmv_band_split_launch) && bp->has_short_group) cut = bp->short_first_block;
        if (short_tiled && bp->has_short_group) end = bp->short_first_block;
        for (uint32_t part = 0; part < 2; ++part) {
            const uint32_t b0 = part ? cut : 0u, nb = part ? end - cut : cut;
            if (!nb) continue;
#define SPRS_COLD(ACCV, POL)                                                                                              \
    hipLaunchKernelGGL((band_cold_kernel<ACCV, POL>), dim3(nb), dim3(CNT), 0, cstream, (const BandPiece *)sc->pieces,          \
                       (const ColdGroup *)bp->groups, bp->ngroups, (const double *)bp->vals_cold,                             \
                       (const uint32_t *)bp->cid_cold, (const double *)sc->xp, x, y, b0)
            const int64_t pol = options().spmv_band_gather;
            if (acc) {
                if (pol == 1) SPRS_COLD(true, 1);
                else if (pol == 2) SPRS_COLD(true, 2);
                else SPRS_COLD(true, 0);
            } else {
                if (pol == 1) SPRS_COLD(false, 1);
                else if (pol == 2) SPRS_COLD(false, 2);
                else SPRS_COLD(false, 0);
            }
#undef SPRS_COLD
            SPRS_TRY_HIP(hipGetLastError());
        }
    }
    if (overlap) SPRS_TRY_HIP(hipEventRecord(sc->join, sc->aux));
    if (bp->hot_wgs) {
        if (bp->hot_threads == 1024)
            hipLaunchKernelGGL((band_hot_kernel<1024, false, false>), dim3(bp->hot_wgs), dim3(1024), 0, stream,
                               (const BandPiece *)sc->pieces, (const uint32_t *)bp->hot_wg_off, bp->nh, bp->G,
                               (const double *)bp->vals_hot, (const void *)bp->cid_hot, (const double *)sc->xp, y);
        else
            hipLaunchKernelGGL((band_hot_kernel<512, false, false>), dim3(bp->hot_wgs), dim3(512), 0, stream,
                               (const BandPiece *)sc->pieces, (const uint32_t *)bp->hot_wg_off, bp->nh, bp->G,
                               (const double *)bp->vals_hot, (const void *)bp->cid_hot, (const double *)sc->xp, y);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (short_tiled) {
        // the short rows with the 8192 hottest x entries in LDS: pieces + npieces = the short piece, a table of one "slice"
        const BandPiece *sp = (const BandPiece *)sc->pieces + bp->npieces;
        if (acc)
            hipLaunchKernelGGL((band_hot_kernel<1024, true, true>), dim3(bp->short_wgs), dim3(1024), 0, stream, sp,
                               (const uint32_t *)bp->short_wg_off, 1u, bp->Gs, (const double *)bp->vals_cold,
                               (const void *)bp->cid_cold, (const double *)sc->xp, y);
        else
            hipLaunchKernelGGL((band_hot_kernel<1024, true, false>), dim3(bp->short_wgs), dim3(1024), 0, stream, sp,
                               (const uint32_t *)bp->short_wg_off, 1u, bp->Gs, (const double *)bp->vals_cold,
                               (const void *)bp->cid_cold, (const double *)sc->xp, y);
        SPRS_TRY_HIP(hipGetLastError());
    }
    if (overlap) SPRS_TRY_HIP(hipStreamWaitEvent(stream, sc->join, 0));
    if (bp->nspills) {
        hipLaunchKernelGGL(band_carry_kernel, dim3((bp->nspills + 255) / 256), dim3(256), 0, stream, (const Spill *)bp->spills,
                           bp->nspills, (const double *)sc->carry, sc->partial, y);
        SPRS_TRY_HIP(hipGetLastError());
    }
    const dim3 rg((bp->n_long + WAVE - 1) / WAVE), rb(RNW * WAVE);
    if (acc)
        hipLaunchKernelGGL(band_reduce_kernel<true>, rg, rb, 0, stream, (const double *)sc->partial,
                           (const unsigned long long *)bp->wmask, (const uint32_t *)bp->wbase,
                           (const uint32_t *)bp->long_rows, y, bp->n_long, bp->npieces, (uint32_t)options().spmv_xmask);
    else
        hipLaunchKernelGGL(band_reduce_kernel<false>, rg, rb, 0, stream, (const double *)sc->partial,
                           (const unsigned long long *)bp->wmask, (const uint32_t *)bp->wbase,
                           (const uint32_t *)bp->long_rows, y, bp->n_long, bp->npieces, (uint32_t)options().spmv_xmask);
    SPRS_TRY_HIP(hipGetLastError());
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
