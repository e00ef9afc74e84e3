// Kernels of one SpMV on the BANDED plan (spmv_band.hip builds the plan and launches them) — device twin of
// prod::mul_acc_mat_vec_csr (sprs/src/sparse/prod.rs:103-127) and of the one-column prod::csr_mulacc_dense_colmaj
// (prod.rs:274-298).  Included by spmv_band.hip only.
//
// Geometry shared by every piece of the plan (hot slices, cold pieces, the short rows):
//   * a WAVE TILE is 512 consecutive entries of a piece; lane l owns the 8 consecutive entries 8 l .. 8 l + 7;
//   * values are stored transposed so that the four coalesced 16-byte loads of a lane return its own entries:
//     entry 8 l + 2 p + e of tile w at vals[512 w + 128 p + 2 l + e];
//   * the row structure travels with the column ids: a flag bit marks the first entry of a row inside the piece,
//     tile_row[w] is the compact row (position in the piece's row list) of the first row starting in tile w;
//   * a RANGE is a run of consecutive tiles walked by ONE wave: the sum of the row that is open at the end of a tile stays
//     in a register and is completed in the next tile, so that only the row open at the START of a range needs a fix-up
//     (band_carry_kernel: one record per range instead of one per tile; round 2 had 630 000 spills per SpMV, now ~25 000);
//   * the sums of the rows that end in a tile are consecutive compact rows: they leave the wave coalesced through a 1 KiB
//     LDS window per wave (band_tile_sums).
#pragma once
#include "spmv_shared.hpp"

namespace sprs_hip {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int WT = 512;                       // entries per wave tile
constexpr int WPASS = WT / (WAVE * 2);        // 16-byte value loads per lane and tile
constexpr int EPL = WT / WAVE;                // entries per lane
constexpr int HOT_THREADS = 1024;             // one workgroup per CU: 16 waves share the x tile
constexpr int HOT_WAVES = HOT_THREADS / WAVE;
constexpr int CNT = 256;                      // threads of a cold / short workgroup: 4 independent waves
constexpr uint32_t ROW_START = 0x8000u;       // flag bit in a hot column id (14 bits of local column below it)
constexpr uint32_t ROW_START32 = 0x80000000u; // flag bit in a cold label
static_assert(WPASS == 4 && EPL == 8, "wave tile geometry");

// One piece of the plan, as the kernels see it.
struct BandPiece {
    const uint32_t *rowidx;     // nr: long-row number (short piece: row of y) of compact row r
    const uint32_t *tile_row;   // ntiles + 1: first compact row starting at / after tile c
    double *out;                // nr partial sums, one per compact row, in row-list order (null for the short piece: it writes y)
    uint64_t ent0;              // first entry of the piece in the value / column-id arrays of its class
    uint64_t nnz;
    uint32_t nr, ntiles;
    uint32_t x0;                // hot: first label of the slice
    uint32_t to_y;              // short piece
    uint32_t range0;            // first range of the piece in the range table (cold / short pieces: ranges of `cold_tiles` tiles)
    uint32_t pad;
};

// A SEGMENT is a run of consecutive tiles of one piece, cut into RANGES of `run` tiles (the last one may be shorter): range r
// of the segment = tiles tile0 + r * run ...; a range is walked by ONE wave.  Hot slices: a segment is the part of a slice
// one workgroup takes (x tile loaded once), its 16 waves take the ranges r = wave, wave + 16, ... — at any time the
// workgroup streams 16 neighbouring ranges, i.e. one contiguous window of the arrays.  (First version of this kernel: one
// long range per wave; the 4096 waves of the chip then streamed 4096 windows a constant 557 056 bytes = 17 x 32 KiB apart
// and hit the same memory channels in step: 909 us against 585 us for the same bytes, profiles/r05b.)
// Cold pieces and the short piece are one segment each.  Ranges are numbered segment after segment: carry[range0 + r].
struct Seg {
    uint32_t piece, tile0, ntiles, range0, run, pad0, pad1, pad2;
};

// What the hot kernel needs of a segment, in ONE 64-byte record (one scalar load): until round 6 a workgroup went
// wg_seg -> segs[s] -> pieces[seg.piece] -> data, three dependent round trips in front of every x tile — nothing beside the
// 2 180 tiles per CU of R-MAT 10M, a third of the time of a small plan whose workgroups hold 50 tiles in two or three segments.
struct alignas(64) HotSeg {
    uint64_t ent0, nnz;                 // of the slice: first entry in the hot arrays, entries
    const uint32_t *tile_row;           // of the slice
    uint64_t out_off;                   // the slice's first partial sum in the scratch's array
    uint32_t x0, tile0, ntiles, range0; // first label of the slice; the segment's tiles; its first range
    uint32_t run, pad0, pad1, pad2;
};
static_assert(sizeof(HotSeg) == 64, "one scalar load");

struct ColdGroup {              // a run of blocks of the cold launch
    uint32_t first_block, first_piece, npieces;   // npieces 8: block b -> piece b % 8 (XCD b % 8); 1: one piece
};

// A pointer loaded from a struct in memory is a generic ("flat") pointer to the compiler, and a flat load counts on the
// LDS counter as well (round 2, found in the ISA): the kernels retype what they load as GLOBAL memory.
#ifdef SPRS_HIP_EMU
#define SPRS_GLOBAL_AS
#define SPRS_HOT_WAVES_ATTR
#else
#define SPRS_GLOBAL_AS __attribute__((address_space(1)))
#define SPRS_HOT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
struct PieceView {
    const SPRS_GLOBAL_AS uint32_t *rowidx, *tile_row;
    SPRS_GLOBAL_AS double *out;
    uint64_t ent0, nnz;
    uint32_t nr, ntiles, x0, to_y, range0;
    __device__ __forceinline__ PieceView(const BandPiece &p)
        : rowidx((const SPRS_GLOBAL_AS uint32_t *)p.rowidx), tile_row((const SPRS_GLOBAL_AS uint32_t *)p.tile_row),
          out((SPRS_GLOBAL_AS double *)p.out), ent0(p.ent0), nnz(p.nnz), nr(p.nr), ntiles(p.ntiles), x0(p.x0), to_y(p.to_y),
          range0(p.range0) {}
};

// ---- wave primitives without the LDS ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t band_scan_incl_u32(uint32_t v, uint32_t lane) {
    (void)lane;
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);     // row_shr:1 (zeros shifted in)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// Segmented inclusive scan over the lanes.  In: S = the sum of the run that is open at the end of the lane, F = 1 when a
// row starts inside the lane.  Out: S = the sum of the run open at the end of the lane INCLUDING what lower lanes hold
// of it; F = 1 when a row starts in this lane or below.  The operator (S, F) o (s, f) = (f ? s : S + s, F | f) is
// associative: the DPP scan pattern of band_scan_incl_u32 applies (lanes that receive nothing get the identity (0, 0)).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ void band_seg_step(double &S, uint32_t &F) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(S), CTRL, ROW_MASK, 0xf, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(S), CTRL, ROW_MASK, 0xf, BOUND);
    const uint32_t fs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)F, CTRL, ROW_MASK, 0xf, BOUND);
    const double vs = __hiloint2double(hi, lo);
    S = F ? S : vs + S;
    F |= fs;
}

__device__ __forceinline__ void band_seg_scan(double &S, uint32_t &F, uint32_t lane) {
    (void)lane;
    band_seg_step<0x111, 0xf, true>(S, F);
    band_seg_step<0x112, 0xf, true>(S, F);
    band_seg_step<0x114, 0xf, true>(S, F);
    band_seg_step<0x118, 0xf, true>(S, F);
    band_seg_step<0x142, 0xa, false>(S, F);
    band_seg_step<0x143, 0xc, false>(S, F);
}

// value of the lane below (lane 0: 0.0, flag 0)
__device__ __forceinline__ double band_lane_below(double S, uint32_t lane) {
    (void)lane;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(S), 0x138, 0xf, 0xf, true);   // wave_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(S), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double band_read_lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), WAVE - 1);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), WAVE - 1);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// Row sums of one wave tile.
//   pr[q]   product of the lane's entry q (0.0 for the padding behind the piece's last entry)
//   fb      bit q = entry q is the first of its row inside the piece
//   R0      compact row of the first row that starts in the tile
//   open    (in / out, wave-uniform) sum of the row that is open when the tile begins / ends
//   mine    (in / out, wave-uniform) a row has started in this range before / by the end of this tile
//   last    (out, wave-uniform) compact row of the last row that has started so far (valid when mine)
// Every row that ENDS inside the tile is emitted: emit(compact row, sum).  The row open at the start of the range began
// in an earlier range: what this range holds of it goes to *carry (band_carry_kernel adds it where it belongs).
// Order of the additions inside a row: entry order inside a lane, lanes in order, tiles in order — fixed by the plan,
// the same in every run (no float atomics anywhere in the SpMV).
// ---------------------------------------------------------------------------------------------
// The sums leave the wave COALESCED: the rows that end inside a tile are consecutive compact rows (R0 - 1 .. R0 + nf - 2),
// so the lanes park their sums in a small LDS window of the wave (STG doubles, by row) and the wave then writes the
// window out, lane = row.  (Stores straight from the lanes that hold the sums — eight masked 8-byte stores per tile at
// scattered addresses — cost the hot kernel 375 us for 0.32 GB of partial sums: 880 against 505 us without them,
// profiles/r05d.)  Tiles with more row ends than the window take several passes.
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave complete in order; this keeps the COMPILER from moving them across the hand-over
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int STG = 128;
constexpr uint32_t hot_lds_bytes(uint32_t xt_log2) { return ((1u << xt_log2) + (uint32_t)HOT_WAVES * 128u) * 8u; }     // doubles of a wave's staging window: hot 16 x 1 KiB + x tile 128 KiB leave 16 KiB of LDS to the cold workgroups

// What band_tile_sums leaves in the wave's LDS window for a LATER flush: the hot kernel writes a tile's sums out after it
// has waited for the next tile's loads — a store issued right before that wait is waited for as well (loads and stores
// share the counter on this chip), which exposed the whole store latency once per tile (profiles/r05f: 140 us).
struct Pending {
    uint32_t cnt = 0;           // sums parked in the window
    uint32_t row0 = 0;          // compact row of the first one
    bool first_to_carry = false;   // the first one belongs to the row that was open when the range began
};

template <typename Out>
__device__ __forceinline__ void band_flush(Pending &pd, uint32_t lane, SPRS_GLOBAL_AS double *carry, const double *stage, Out out) {
    for (uint32_t j = lane; j < pd.cnt; j += WAVE) {
        const double v = stage[j];
        if (j == 0 && pd.first_to_carry) *carry = v;
        else out(pd.row0 + j, v);
    }
    wave_lds_fence();                                                   // read before the next window / tile overwrites it
    pd.cnt = 0;
}

// DEFER: the last window of the tile stays parked (pd) for band_flush; otherwise everything is written out here.
template <bool DEFER, typename Out>
__device__ __forceinline__ void band_tile_sums(const double (&pr)[EPL], uint32_t fb, uint32_t lane, uint32_t R0, double &open, bool &mine,
                                               uint32_t &last, SPRS_GLOBAL_AS double *carry, double *stage, Pending &pd, Out out) {
    const uint32_t nfl = (uint32_t)__popc(fb);                          // rows starting in this lane
    const uint32_t incl = band_scan_incl_u32(nfl, lane);
    const uint32_t prefix = incl - nfl;                                 // rows starting in lower lanes
    const uint32_t nf = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    // ---- serial fold of the lane's entries: ev[q] = sum of the run that ends at the row start of entry q ------------
    double ev[EPL];
    double run = 0.0, head = 0.0;
    uint32_t seen = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        ev[q] = run;                                                    // (only read where entry q starts a row)
        if ((fb >> q) & 1u) {
            if (seen == 0) head = run;                                  // the run that was open when the lane began ends here
            run = 0.0;
            ++seen;
        }
        run += pr[q];
    }
    // ---- runs that cross lanes -------------------------------------------------------------------------------------
    double S = run;
    uint32_t F = seen ? 1u : 0u;
    band_seg_scan(S, F, lane);
    double before = band_lane_below(S, lane);                           // open run at the end of the lane below
    if (prefix == 0) before = open + before;                            // no row has started in the tile so far: the range's open row
    const double vhead = before + head;                                 // the run that ends at this lane's first row start
    // ---- out, window by window: row end number o of the tile (o = 0: the range's open row) is compact row R0 - 1 + o ----
    for (uint32_t lo = 0; lo < nf; lo += STG) {                         // wave-uniform
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            if ((fb >> q) & 1u) {
                const uint32_t below = fb & ((1u << q) - 1u);
                const uint32_t o = prefix + (uint32_t)__popc(below) - lo;
                if (o < (uint32_t)STG) stage[o] = below ? ev[q] : vhead;
            }
        }
        wave_lds_fence();
        pd.cnt = nf - lo < (uint32_t)STG ? nf - lo : (uint32_t)STG;
        pd.row0 = R0 - 1u + lo;
        pd.first_to_carry = lo == 0 && !mine;                           // ... began before the range
        if (!DEFER || lo + STG < nf) band_flush(pd, lane, carry, stage, out);
    }
    const double s63 = band_read_lane63(S);
    open = nf ? s63 : open + s63;
    if (nf) {
        mine = true;
        last = R0 + nf - 1;
    }
}

// ---------------------------------------------------------------------------------------------
// hot slices: x tile in LDS, 16-bit local column ids.
//
// One workgroup per CU (16 waves, the x tile takes 64 or 128 KiB of the 160 KiB of LDS), `rounds` workgroups per CU in
// all.  A workgroup owns an equal share of the wave tiles of ALL hot slices, i.e. a few SEGMENTS (parts of a slice): per
// segment it loads the slice's x tile once, then its 16 waves walk their ranges independently — no barrier until the
// next segment.  (Round 2 launched one workgroup per 131 072 entries: 2 146 workgroups of unequal slices, 8.4 rounds
// on 256 CUs and a tail of ~7 %; with equal shares every CU streams until the end.)
// ---------------------------------------------------------------------------------------------
// 96 VGPRs: the 4 waves per SIMD of this kernel then leave room for TWO 64-register waves of the gather kernels beside them
// (at 98 it was one, and the cold launch crawled beside the hot one: 370 us instead of 86 alone, profiles/r05h)
template <bool ACC, bool TOY>
__device__ __forceinline__ void band_cold_wave(const PieceView d, uint32_t r, uint32_t ct, const double *__restrict__ vals,
                                               const uint32_t *__restrict__ cid, const double *__restrict__ xp, double *__restrict__ y,
                                               double *__restrict__ carry, double *stage, uint32_t lane);

// (Round 6 also ran the short rows in THIS kernel's prologue — the upper 8 waves of a workgroup walking short-row ranges while the lower
// 8 fetched the first x tile: the short tile's chain of round trips, 13 us, then stood in front of every workgroup's first hot tile:
// hot kernel 36.7 -> 45.7 us for a reduction alone of 12.4 instead of a 19.3 us tail, 0.0696 - 0.0702 against 0.0684 - 0.0689 ms on
// R-MAT 1M, profiles/r15o.  Not kept.)
template <int XT_LOG2>
__global__ __launch_bounds__(HOT_THREADS) SPRS_HOT_WAVES_ATTR void band_hot_kernel(const HotSeg *__restrict__ hsegs,
                                                               const HotSeg *__restrict__ wg_first, const double *__restrict__ vals,
                                                               const uint16_t *__restrict__ cid, const double *__restrict__ xp,
                                                               double *__restrict__ partial, double *__restrict__ carry, uint32_t dbg,
                                                               unsigned long long *__restrict__ prof, uint32_t xcd_shares) {
    constexpr int XT = 1 << XT_LOG2;
    // developer builds, option spmv_band_debug & 16 (env SPRS_HIP_HOTPROF): when does each workgroup start, see its first x tile,
    // and end (100 MHz wall clock) — is the hot kernel's time its work or its tail?
    if constexpr (DEVTOOLS) {
        if (prof && threadIdx.x == 0) prof[3 * blockIdx.x] = (unsigned long long)wall_clock64();
    }
    // dynamic LDS (hot_lds_bytes): with the size known at compile time the compiler sees that only 4 waves per SIMD fit and
    // spends up to 128 registers; it is asked for 5 (96 registers) so that two gather waves fit beside each hot wave
#ifdef SPRS_HIP_EMU
    static double lds[XT + HOT_WAVES * STG];
#else
    extern __shared__ __attribute__((aligned(16))) double lds[];
#endif
    double *xs = lds;                                                    // XT doubles: the x tile
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    double *stage = lds + XT + wave * STG;                              // the wave's window for outgoing sums
    // share of this workgroup.  xcd_shares: block b runs on XCD b % 8 (observed dispatch order; only speed depends on it) and every
    // XCD takes a CONTIGUOUS run of shares — the workgroups that stream one slice then sit behind one L2, which serves its x tile
    // to all but the first of them (dealt round-robin, every XCD's L2 fetched every slice: 256 workgroups x 128 KiB = 32 MB through
    // the fabric in the first microseconds of a small plan's hot kernel).
    uint32_t share = blockIdx.x;
    if (xcd_shares) {
        const uint32_t nq = gridDim.x >> 3, rem = gridDim.x & 7u, k = blockIdx.x & 7u;
        share = k * nq + (k < rem ? k : rem) + (blockIdx.x >> 3);
    }
    // the share's FIRST segment comes with its segment range in one record (wg_first[share]: pad0 / pad1 = first segment, end) —
    // one round trip in front of the first x tile instead of two (wg_seg -> hsegs)
    HotSeg nxt = wg_first[share];
    const uint32_t s0 = nxt.pad0, s1 = nxt.pad1;
    for (uint32_t s = s0; s < s1; ++s) {
        const HotSeg seg = nxt;
        if (s + 1 < s1) nxt = hsegs[s + 1];                              // (requested a whole segment before it is needed)
        struct {
            const SPRS_GLOBAL_AS uint32_t *tile_row;
            SPRS_GLOBAL_AS double *out;
            uint64_t ent0, nnz;
            uint32_t x0;
        } d{(const SPRS_GLOBAL_AS uint32_t *)seg.tile_row, (SPRS_GLOBAL_AS double *)(partial + seg.out_off), seg.ent0, seg.nnz, seg.x0};
        if (s != s0) __syncthreads();                                    // every wave is done with the previous x tile
        {   // x tile of the slice -> LDS (xp is padded to a whole number of tiles)
            const dbl2 *src = (const dbl2 *)(xp + d.x0);
            dbl2 v[XT / (2 * HOT_THREADS)];
#pragma unroll
            for (int q = 0; q < XT / (2 * HOT_THREADS); ++q) v[q] = src[q * HOT_THREADS + tid];
#pragma unroll
            for (int q = 0; q < XT / (2 * HOT_THREADS); ++q) *(dbl2 *)&xs[2 * (q * HOT_THREADS + tid)] = v[q];
        }
        // the wave's tiles, in the order it walks them: ranges r = wave, wave + 16, ...; inside a range tile after tile
        const uint32_t run = seg.run, n = seg.ntiles;
        dbl2 av[WPASS];
        u32x4 cw = {0u, 0u, 0u, 0u};
        uint32_t R0n = 0;
        auto request = [&](uint32_t w) {
            const uint64_t g = d.ent0 + (uint64_t)w * WT;
#pragma unroll
            for (int p = 0; p < WPASS; ++p) av[p] = __builtin_nontemporal_load((const dbl2 *)(vals + g + p * (WAVE * 2) + lane * 2));
            cw = __builtin_nontemporal_load((const u32x4 *)(cid + g + lane * EPL));
            R0n = d.tile_row[w];
        };
        uint32_t r = wave;                                               // current range
        uint32_t t = r * run;                                            // current tile, relative to the segment
        if (t < n) request(seg.tile0 + t);                               // the first tile is requested before the barrier
        __syncthreads();                                                 // xs complete
        if constexpr (DEVTOOLS) {
            if (prof && threadIdx.x == 0 && s == s0) prof[3 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
        }
        double open = 0.0;
        bool mine = false;
        uint32_t last = 0;
        Pending pd;
        SPRS_GLOBAL_AS double *pd_slot = (SPRS_GLOBAL_AS double *)carry;
        while (t < n) {                                                  // wave-uniform
            const uint32_t w = seg.tile0 + t;
            const uint64_t base = (uint64_t)w * WT;
            const uint32_t cnt = d.nnz - base < (uint64_t)WT ? (uint32_t)(d.nnz - base) : (uint32_t)WT;
            double pr[EPL], xv[EPL];
            uint32_t fb = 0;
            // all eight LDS gathers are issued before the first product needs one (a conditional read per entry compiles to
            // eight branches with a full LDS wait each)
#pragma unroll
            for (int p = 0; p < WPASS; ++p) {
                const uint32_t c2 = cw[p];
                xv[2 * p] = xs[c2 & (XT - 1)];
                xv[2 * p + 1] = xs[(c2 >> 16) & (XT - 1)];
                fb |= ((c2 >> 15) & 1u) << (2 * p);
                fb |= ((c2 >> 31) & 1u) << (2 * p + 1);
            }
#pragma unroll
            for (int p = 0; p < WPASS; ++p) {
                pr[2 * p] = av[p][0] * xv[2 * p];
                pr[2 * p + 1] = av[p][1] * xv[2 * p + 1];
            }
            if (cnt < (uint32_t)WT) {                                    // last tile of a slice (wave-uniform): the padding (value 0, id 0) must not
#pragma unroll                                                           // turn an infinite x[first label] into a NaN
                for (int q = 0; q < EPL; ++q) pr[q] = lane * EPL + q < cnt ? pr[q] : 0.0;
            }
            // tile_row[w] is the LAST load of the tile's request: taken here, in front of the flush, so that its wait covers only
            // loads issued a whole tile ago.  (Until round 5 the compiler sank this copy behind the flush's stores, and on gfx950
            // loads and stores count on ONE in-order vmcnt: the wave then waited for the acknowledgement of its stores with no
            // load in flight, once per tile — `s_waitcnt vmcnt(0)` between the flush and the next request in the ISA; the probe
            // scripts/probes/stream_store.hip prices such a drain at 1.0 - 2.3 us per tile.)
            uint32_t R0 = R0n;
#ifndef SPRS_HIP_EMU
            asm volatile("" : "+v"(R0)::"memory");
#endif
            // the tile after this one: the next of the range, or the first of the wave's next range
            const uint32_t rend = (r + 1) * run < n ? (r + 1) * run : n;
            const bool range_ends = t + 1 >= rend;
            const uint32_t tn = range_ends ? (r + HOT_WAVES) * run : t + 1;
            SPRS_GLOBAL_AS double *cslot = (SPRS_GLOBAL_AS double *)carry + seg.range0 + r;
            auto out = [&](uint32_t row, double v) {
                if (DEVTOOLS && (dbg & 1u)) return;                      // no stores of the row sums
                __builtin_nontemporal_store(v, &d.out[row]);              // read again by the reduction, a few GB of traffic later

            };
            if (pd.cnt) band_flush(pd, lane, pd_slot, stage, out);        // the previous tile's sums, before this tile's loads are asked for
            if (tn < n) request(seg.tile0 + tn);                         // streams while this tile is summed
            if constexpr (DEVTOOLS) {                                    // timing experiments (option spmv_band_debug): WRONG results
                if (dbg & 2u) fb &= 0x01u;                               // at most one row start per lane
                if (dbg & 4u) fb = 0;                                    // no row starts at all
            }
            if (DEVTOOLS && (dbg & 8u)) {
                band_tile_sums<false>(pr, fb, lane, R0, open, mine, last, cslot, stage, pd, out);
            } else {
                band_tile_sums<true>(pr, fb, lane, R0, open, mine, last, cslot, stage, pd, out);
                pd_slot = cslot;
            }
            if (range_ends) {
                if (lane == 0) {
                    if (mine) d.out[last] = open;                        // the row still open at the end of the range: its sum so far
                    else *cslot = open;                                  // no row starts in the whole range: all of it belongs to an earlier row
                }
                open = 0.0;
                mine = false;
                r += HOT_WAVES;
            }
            t = tn;
        }
        if (pd.cnt) band_flush(pd, lane, pd_slot, stage, [&](uint32_t row, double v) {
            if (DEVTOOLS && (dbg & 1u)) return;
            d.out[row] = v;
        });
    }
    if constexpr (DEVTOOLS) {
        if (prof) {
            __syncthreads();
            if (threadIdx.x == 0) prof[3 * blockIdx.x + 2] = (unsigned long long)wall_clock64();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cold pieces and the short rows: x gathered through L1 / L2 from xp, 32-bit labels.
//
// Same wave-tile scheme without an x tile and without LDS: four independent waves per workgroup, each walking one range
// of `ct` consecutive tiles, so that many workgroups share a CU beside the hot kernel's one and thousands of gathers are
// in flight per CU.  Labels are stored transposed for two coalesced 16-byte loads: entry 8 l + 4 p + e at
// cid[512 w + 256 p + 4 l + e].  Pieces start at multiples of 512 entries and are padded with (label 0, value 0).
// The short piece (to_y) writes y[rowidx[r]] instead of a partial sum.
// ---------------------------------------------------------------------------------------------
// 8 waves per SIMD (64 VGPRs, 20 bytes of scratch): the gathers live on the number of waves in flight, and two such waves fit
// beside each wave of the hot kernel.  (Non-temporal and device-scope gathers were measured slower in round 2.)
struct ColdArgs {
    const BandPiece *pieces;
    const ColdGroup *groups;
    uint32_t ngroups;
    const double *vals;
    const uint32_t *cid;
    const double *xp;
    double *y, *carry;
    uint32_t block0, ct;
    uint32_t direct;             // 1: the launch walks ONE piece, given by value below (no look-up of group and piece in memory:
    BandPiece piece;             //    two dependent round trips less in front of the first tile — the short rows of a small plan)
};

template <bool ACC, bool TOY>
__device__ __forceinline__ void band_cold_body(const ColdArgs &ca, uint32_t block, double (*stage_s)[STG]) {
    constexpr int WPB = CNT / WAVE;
    const BandPiece *__restrict__ pieces = ca.pieces;
    const ColdGroup *__restrict__ groups = ca.groups;
    const uint32_t ngroups = ca.ngroups, ct = ca.ct;
    const double *__restrict__ vals = ca.vals;
    const uint32_t *__restrict__ cid = ca.cid;
    const double *__restrict__ xp = ca.xp;
    double *__restrict__ y = ca.y;
    double *__restrict__ carry = ca.carry;
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    double *stage = stage_s[wave];
    const uint32_t bid = block + ca.block0;
    uint32_t r;                                                          // range of the piece (= of its segment)
    BandPiece pc;
    if (ca.direct) {
        pc = ca.piece;
        r = block * WPB + wave;
    } else {
        uint32_t g = 0;
        while (g + 1 < ngroups && bid >= groups[g + 1].first_block) ++g;     // block-uniform
        const ColdGroup cg = groups[g];
        const uint32_t lb = bid - cg.first_block;
        const uint32_t pi = cg.first_piece + (cg.npieces == 1 ? 0u : (lb & 7u));
        r = (cg.npieces == 1 ? lb : (lb >> 3)) * WPB + wave;
        pc = pieces[pi];
    }
    band_cold_wave<ACC, TOY>(PieceView(pc), r, ct, vals, cid, xp, y, carry, stage, lane);
}

// one wave, one range (ct consecutive tiles) of a gather piece
template <bool ACC, bool TOY>
__device__ __forceinline__ void band_cold_wave(const PieceView d, uint32_t r, uint32_t ct, const double *__restrict__ vals,
                                               const uint32_t *__restrict__ cid, const double *__restrict__ xp, double *__restrict__ y,
                                               double *__restrict__ carry, double *stage, uint32_t lane) {
    const uint32_t t0 = r * ct;
    if (t0 >= d.ntiles) return;                                          // wave-uniform; no workgroup barrier below
    const uint32_t tend = t0 + ct < d.ntiles ? t0 + ct : d.ntiles;
    dbl2 av[WPASS];
    u32x4 lw[2];
    uint32_t R0n = 0;
    auto request = [&](uint32_t w) {
        const uint64_t gpos = d.ent0 + (uint64_t)w * WT;
#pragma unroll
        for (int p = 0; p < 2; ++p) lw[p] = __builtin_nontemporal_load((const u32x4 *)(cid + gpos + p * (WAVE * 4) + lane * 4));
#pragma unroll
        for (int p = 0; p < WPASS; ++p) av[p] = __builtin_nontemporal_load((const dbl2 *)(vals + gpos + p * (WAVE * 2) + lane * 2));
        R0n = d.tile_row[w];
    };
    request(t0);
    double open = 0.0;
    bool mine = false;
    uint32_t last = 0;
    SPRS_GLOBAL_AS double *cslot = (SPRS_GLOBAL_AS double *)carry + d.range0 + r;
    // Where a short row's sum goes (rowidx) is known as soon as the tile's first row is: the rows that end in a tile are the compact
    // rows R0 - 1 + o, o = lane + 64 m — requested for m < 4 (256 row ends; R-MAT's short rows: ~195 per tile) beside the gathers
    // instead of one dependent load per flush pass (a short tile was a chain of six round trips, four of them these: 13 us on
    // R-MAT 1M whichever launch it ran in, profiles/r15n).
    uint32_t yr_pre[4] = {0u, 0u, 0u, 0u}, yr_base = 0;
    auto emit_y = [&](uint32_t row, double v) {
        const uint32_t m = (row - yr_base) >> 6;                         // wave-uniform inside a flush pass; lane 0's last row: whatever
        uint32_t yr;
        if (TOY && m < 4u && ((row - yr_base) & 63u) == lane) yr = m == 0 ? yr_pre[0] : m == 1 ? yr_pre[1] : m == 2 ? yr_pre[2] : yr_pre[3];
        else yr = d.rowidx[row];
        if constexpr (ACC) y[yr] = y[yr] + v;                            // every compact row has entries: empty rows are never touched (prod.rs:120-126)
        else y[yr] = v;
    };
    auto emit_p = [&](uint32_t row, double v) { d.out[row] = v; };
    for (uint32_t w = t0; w < tend; ++w) {
        const uint64_t base = (uint64_t)w * WT;
        const uint32_t cnt = d.nnz - base < (uint64_t)WT ? (uint32_t)(d.nnz - base) : (uint32_t)WT;
        double xv[EPL];
        uint32_t fb = 0;
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const uint32_t c = lw[q / 4][q % 4];
            xv[q] = xp[c & ~ROW_START32];                                // padding: label 0, value 0, never summed into a row
            fb |= (c >> 31) << q;
        }
        if constexpr (TOY) {
            yr_base = R0n - 1u;                                          // (tile 0 of the piece: R0n = 0, the wrap only names the range's open row, which has no y)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t rr = yr_base + lane + 64u * (uint32_t)m;
                yr_pre[m] = d.rowidx[rr < d.nr ? rr : d.nr - 1u];
            }
        }
        double pr[EPL];
#pragma unroll
        for (int q = 0; q < EPL; ++q) pr[q] = av[q / 2][q % 2] * xv[q];
        if (cnt < (uint32_t)WT) {                                        // last tile of the piece (wave-uniform): see band_hot_kernel
#pragma unroll
            for (int q = 0; q < EPL; ++q) pr[q] = lane * EPL + q < cnt ? pr[q] : 0.0;
        }
        // (no prefetch of the next tile here: its 24 registers would halve the waves per SIMD, and the gathers live on those)
        Pending pd;
        if constexpr (TOY) band_tile_sums<false>(pr, fb, lane, R0n, open, mine, last, cslot, stage, pd, emit_y);
        else band_tile_sums<false>(pr, fb, lane, R0n, open, mine, last, cslot, stage, pd, emit_p);
        if (w + 1 < tend) request(w + 1);
    }
    if (lane == 0) {
        if (!mine) *cslot = open;
        else if constexpr (TOY) emit_y(last, open);
        else emit_p(last, open);
    }
}

template <bool ACC, bool TOY>
__global__ __launch_bounds__(CNT, 8) void band_cold_kernel(ColdArgs ca) {
    __shared__ __attribute__((aligned(16))) double stage_s[CNT / WAVE][STG];
    band_cold_body<ACC, TOY>(ca, blockIdx.x, stage_s);
}

// ---------------------------------------------------------------------------------------------
// The row that is open at the START of a range began in an earlier range: the part of it each range holds (its HEAD) was
// left in carry[range].  The first range of a run of ranges that continue the same row adds their heads, in range order,
// to that row's sum (a (row, piece) partial, or y for the short rows).  Which ranges have a head, and whose, is read off the
// plan (the first entry's row-start flag, tile_row) ONCE, when the plan is built: bp_spill_kernel leaves one record per run.
// ---------------------------------------------------------------------------------------------
struct RangeRef {
    uint32_t piece, tile0;
    bool valid;
};

__device__ __forceinline__ RangeRef range_of(const Seg *__restrict__ segs, uint32_t nsegs, uint32_t i) {
    uint32_t lo = 0, hi = nsegs;                                         // last segment with range0 <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].range0 <= i) lo = mid;
        else hi = mid;
    }
    const Seg sg = segs[lo];
    const uint32_t t = (i - sg.range0) * sg.run;
    return RangeRef{sg.piece, sg.tile0 + t, t < sg.ntiles};
}

__device__ __forceinline__ bool range_has_head(const BandPiece &d, uint32_t piece, uint32_t tile0, const uint16_t *__restrict__ cid_hot,
                                               const uint32_t *__restrict__ cid_cold, uint32_t nhot) {
    const uint64_t e = d.ent0 + (uint64_t)tile0 * WT;                    // entry 0 of lane 0 sits first in both layouts
    return piece < nhot ? !(cid_hot[e] & ROW_START) : !(cid_cold[e] & ROW_START32);
}

// One record per run of ranges that continue one row, found once when the plan is built.
struct Spill {
    uint64_t dst;               // index into the partial sums, or row of y for the short piece
    uint32_t first, n;          // carry slots first .. first + n - 1
    uint32_t to_y, j;           // j: long-row number of the destination (partial-sum records)
};

// The long rows' records as the REDUCTION reads them (round 6): sorted by (long row, first carry slot) on the host when the plan
// is built, one offset per block of 64 long rows — the wave that sums a row block adds its rows' heads itself, so that no
// launch stands between the hot slices and the reduction (band_carry_kernel<false> was 5 - 12 us of latency on that path).
struct RSpill {
    uint32_t j, first, n, pad;
};

__global__ __launch_bounds__(256) void bp_spill_kernel(const Seg *__restrict__ segs, uint32_t nsegs, uint32_t nranges,
                                                       const BandPiece *__restrict__ pieces, const uint64_t *__restrict__ pair_off,
                                                       uint32_t nhot, const uint16_t *__restrict__ cid_hot,
                                                       const uint32_t *__restrict__ cid_cold,
                                                       Spill *__restrict__ spills_y, Spill *__restrict__ spills_partial,
                                                       unsigned int *__restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nranges) return;
    const RangeRef rg = range_of(segs, nsegs, i);
    if (!rg.valid) return;
    const BandPiece d = pieces[rg.piece];
    if (!range_has_head(d, rg.piece, rg.tile0, cid_hot, cid_cold, nhot)) return;
    const uint32_t row = d.tile_row[rg.tile0] - 1;                        // the row open at the start of the range (a piece begins with a row start)
    // a leader has no predecessor that continues the same row (ranges of one piece follow each other in tile order)
    if (i > 0) {
        const RangeRef pj = range_of(segs, nsegs, i - 1);
        if (pj.valid && pj.piece == rg.piece && d.tile_row[pj.tile0] - 1 == row && range_has_head(d, pj.piece, pj.tile0, cid_hot, cid_cold, nhot))
            return;
    }
    uint32_t n = 1;
    for (uint32_t j = i + 1; j < nranges; ++j, ++n) {
        const RangeRef nj = range_of(segs, nsegs, j);
        if (!nj.valid || nj.piece != rg.piece || d.tile_row[nj.tile0] - 1 != row || !range_has_head(d, nj.piece, nj.tile0, cid_hot, cid_cold, nhot))
            break;
    }
    // two lists: the short rows' (into y, applied behind them on their stream) and the long rows' (into the partial sums, in
    // front of the reduction)
    const bool first = d.to_y != 0;
    const unsigned int slot = atomicAdd(count + (first ? 0 : 1), 1u);
    (first ? spills_y : spills_partial)[slot] = Spill{d.to_y ? (uint64_t)d.rowidx[row] : pair_off[rg.piece] + row, i, n, d.to_y, d.rowidx[row]};
}

// per SpMV: the heads of a record's ranges are added to the row's sum in range order.  Two launches: the short rows' records
// (TO_Y: into y, behind the short rows on their stream) and the long rows' (into the partial sums, in front of the reduction).
template <bool TO_Y>
__global__ __launch_bounds__(256) void band_carry_kernel(const Spill *__restrict__ spills, uint32_t nspills, const double *__restrict__ carry,
                                                         double *__restrict__ partial, double *__restrict__ y) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nspills) return;
    const Spill sp = spills[i];
    double *dst = (TO_Y ? y : partial) + sp.dst;
    double acc = *dst;
    for (uint32_t k = 0; k < sp.n; ++k) acc += carry[sp.first + k];
    *dst = acc;
}

// x in the plan's labelling: xp[l] = x[inv[l]] for the labels l0 <= l < l1, four per thread (one 16-byte load of inv, four
// gathers, two 16-byte stores), and y cleared, four rows per thread.  Only the labels of REFERENCED columns are moved (the
// columns nobody references are the last class of the labelling: on R-MAT 47 % of them, in a rank's block of an 8-way cut most;
// xp behind them stays 0.0 and is never read).  Until round 6 this was a scatter over all columns (xp[label[c]] = x[c]: eight-byte
// stores to ten million places) with only the hot labels gathered; the gather reads x in ascending order inside a popularity
// class and writes whole lines.  The hot kernel needs only the hot labels: big plans gather those first on the main stream and
// the rest (and the clearing of y) on the second stream beside it.
// Measured against the scatter below in one session (gpurun_out/r15u): a rank's block of the 8-way cut of R-MAT 10M (a quarter of the
// columns referenced) 0.208 - 0.216 against 0.228 - 0.231 ms; the whole R-MAT 10M (53 % referenced) 1.08 - 1.11 against 1.04 - 1.07 ms — the
// gathers of x cost the fabric more lines than the scatter's stores once most columns take part; R-MAT 1M equal.  So: the gather
// when at most a third of the columns are referenced, the scatter over all columns otherwise.
// (CLEAR_Y tells the two launches of a big plan apart in a profile: <false> the hot labels, <true> the rest and y)
template <bool CLEAR_Y>
__global__ __launch_bounds__(256) void band_gather_kernel(const double *__restrict__ x, const uint32_t *__restrict__ inv, uint32_t l0,
                                                          uint32_t l1, double *__restrict__ xp, double *__restrict__ y_zero, uint64_t rows) {
    const uint64_t t4 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint64_t l = (uint64_t)l0 + t4;
    if (l + 4 <= (uint64_t)l1 && (l0 & 3u) == 0) {
        const u32x4 j = *(const u32x4 *)(inv + l);
        const double a0 = x[j[0]], a1 = x[j[1]], a2 = x[j[2]], a3 = x[j[3]];
        *(dbl2 *)(xp + l) = dbl2{a0, a1};
        *(dbl2 *)(xp + l + 2) = dbl2{a2, a3};
    } else {
        for (uint64_t q = l; q < (uint64_t)l1 && q < l + 4; ++q) xp[q] = x[inv[q]];
    }
    if (CLEAR_Y && y_zero) {
        if (t4 + 4 <= rows && ((uintptr_t)y_zero & 15) == 0) {
            *(dbl2 *)(y_zero + t4) = dbl2{0.0, 0.0};
            *(dbl2 *)(y_zero + t4 + 2) = dbl2{0.0, 0.0};
        } else {
            for (uint64_t r = t4; r < rows && r < t4 + 4; ++r) y_zero[r] = 0.0;
        }
    }
}

// The scatter form: xp[label[c]] = x[c] for the labels in [first_label, nref) — the labels below were gathered (hot labels of a big
// plan); the caller passes nref = every label — and y cleared: four consecutive columns / rows per thread
// (16-byte loads; one column per thread left this kernel latency-bound beside the hot kernel: 360 us instead of 50, profiles/r05e)
__global__ __launch_bounds__(256) void band_permute_kernel(const double *__restrict__ x, const uint32_t *__restrict__ perm,
                                                           uint64_t cols, double *__restrict__ xp, double *__restrict__ y_zero,
                                                           uint64_t rows, uint32_t first_label, uint32_t nref) {
    const uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint32_t span = nref - first_label;                            // label in range <=> label - first_label < span
    if (j + 4 <= cols && (((uintptr_t)x | (uintptr_t)perm) & 15) == 0) {
        const u32x4 l = *(const u32x4 *)(perm + j);
        const dbl2 a = *(const dbl2 *)(x + j), b = *(const dbl2 *)(x + j + 2);
        if (l[0] - first_label < span) xp[l[0]] = a[0];
        if (l[1] - first_label < span) xp[l[1]] = a[1];
        if (l[2] - first_label < span) xp[l[2]] = b[0];
        if (l[3] - first_label < span) xp[l[3]] = b[1];
    } else {
        for (uint64_t c = j; c < cols && c < j + 4; ++c) {
            const uint32_t l = perm[c];
            if (l - first_label < span) xp[l] = x[c];
        }
    }
    if (y_zero) {
        if (j + 4 <= rows && ((uintptr_t)y_zero & 15) == 0) {
            *(dbl2 *)(y_zero + j) = dbl2{0.0, 0.0};
            *(dbl2 *)(y_zero + j + 2) = dbl2{0.0, 0.0};
        } else {
            for (uint64_t r = j; r < rows && r < j + 4; ++r) y_zero[r] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// y[long_rows[j]] (+)= sum of the row's partials, pieces in ascending order (a fixed order: deterministic).
// One WAVE per 64 consecutive long rows (lane = row), no LDS, no barrier.  For piece k, wmask tells which of the 64 rows
// have a partial there and wbase where the first of them sits; the present rows' partials follow each other in memory.
// The table rows of a row block are contiguous over k: one coalesced load puts 64 pieces into the lanes, readlane hands
// them out.  Round 2 split the pieces of a row block over the 4 waves of a workgroup and combined through LDS behind a
// barrier — each wave was two dependent round trips long and the kernel ran at half the bandwidth of its traffic.
// ---------------------------------------------------------------------------------------------
constexpr int RU = 16;       // partials in flight per lane; the table rows are padded to a multiple of it (absent pieces: mask 0)

// lane's bit of a wave-uniform mask as a condition, and the number of mask bits below the lane
__device__ __forceinline__ bool band_mask_bit(uint32_t m_lo, uint32_t m_hi, uint32_t lane) {
    (void)lane;
    return __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)m_hi << 32) | m_lo);     // the mask becomes EXEC / VCC as it is
}
__device__ __forceinline__ uint32_t band_mask_rank(uint32_t m_lo, uint32_t m_hi, uint32_t lane) {
    (void)lane;
    return __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
}

// All pieces of every row, in ascending order.
// (Round 5 tried the transposed form — lane = PARTIAL: a wave owns 128 / 256 long rows with one LDS accumulator each, the
// partials of the block inside a piece are a contiguous run cut into chunks of 64, loaded with every lane busy together with
// one byte per pair that names the row, added with ds_add_f64 — 4 x fewer, full load instructions: 126 - 134 us against
// 122 us, and 99 us even with the partial array read front to back and neither adds nor stores (profiles/r13h ... r13k).  The
// reduction is not bound by its instruction count or its access pattern; the form with the smaller tables stayed.
// Round 6 tried TWO lanes per row for small plans (a wave per 32 rows, lane l / l + 32 summing the lower / upper 32 pieces of a
// chunk, half the dependent groups per wave, twice the waves): R-MAT 1M 21.6 us against 15.4 us alone, 28.6 against 21.7 us
// inside the tail launch (profiles/r15l) — not a chain of round trips either.)
struct ReduceArgs {
    const double *partial;
    const unsigned long long *wmask;
    const uint32_t *wbase, *long_rows;
    const RSpill *rsp;           // the heads of rows that run on from one range into the next (sorted by long row), or null
    const uint32_t *rsp_off;     // nwb + 1: the records of row block wb are rsp[rsp_off[wb]] .. rsp[rsp_off[wb + 1] - 1]
    const double *carry;
    double *y;
    uint32_t n_long, np_pad, nwb;
};

template <bool ACC>
__device__ __forceinline__ void band_reduce_body(const ReduceArgs &ra, uint32_t block, uint32_t nblocks) {
    const double *__restrict__ partial = ra.partial;
    const unsigned long long *__restrict__ wmask = ra.wmask;
    const uint32_t *__restrict__ wbase = ra.wbase;
    const uint32_t *__restrict__ long_rows = ra.long_rows;
    double *__restrict__ y = ra.y;
    const uint32_t n_long = ra.n_long, np_pad = ra.np_pad, nwb = ra.nwb;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    // wave -> row block: block b runs on XCD b % 8 (observed; only speed depends on it): every XCD gets a CONTIGUOUS range
    // of row blocks (neighbouring row blocks read neighbouring partials of every piece, often the same 128-byte line)
    const uint32_t wpb = blockDim.x / WAVE;
    const uint32_t xcd = block & 7u, inx = (block >> 3) * wpb + threadIdx.x / WAVE;   // wave number inside the XCD
    const uint32_t per_xcd = (nwb + 7u) / 8u, waves_per_xcd = (nblocks >> 3) * wpb;
    for (uint32_t i = inx; i < per_xcd; i += waves_per_xcd) {            // wave-uniform
        const uint32_t wb = xcd * per_xcd + i;
        if (wb >= nwb) break;
        const uint64_t j = (uint64_t)wb * WAVE + lane;
        const uint64_t jc = j < n_long ? j : n_long - 1;
        const uint32_t r = long_rows[jc];                               // (requested early: needed only at the very end)
        // The heads of this block's rows that run on from one range into the next (typically 5 - 10 records per block, one carry
        // each) do not depend on the partial sums: record and first carry are requested NOW, beside the tables, and consumed at
        // the end (a small plan's reduction is a chain of dependent round trips, not a stream: every one taken out counts).
        uint32_t sb = 0, se = 0;
        if (ra.rsp_off) {
            sb = ra.rsp_off[wb];
            se = ra.rsp_off[wb + 1];
        }
        uint32_t h_jl = 0xFFFFFFFFu, h_first = 0, h_n = 0;
        double h_c = 0.0;
        if (sb + lane < se) {
            const RSpill rec = ra.rsp[sb + lane];
            h_jl = rec.j & (WAVE - 1);
            h_first = rec.first;
            h_n = rec.n;
            h_c = ra.carry[h_first];                                    // n >= 1
        }
        double s = 0.0;
        const unsigned long long *mrow = wmask + (uint64_t)wb * np_pad;
        const uint32_t *brow = wbase + (uint64_t)wb * np_pad;
        unsigned long long mk_n = lane < np_pad ? mrow[lane] : 0ull;     // lane l: the table row of piece l (np_pad is a multiple of 16)
        uint32_t bs_n = lane < np_pad ? brow[lane] : 0u;
        for (uint32_t k0 = 0; k0 < np_pad; k0 += WAVE) {
            const unsigned long long mk = mk_n;
            const uint32_t bs = bs_n;
            if (k0 + WAVE < np_pad) {                                    // the next 64 pieces' table rows: requested in front of this chunk's sums
                const bool in = k0 + WAVE + lane < np_pad;
                mk_n = in ? mrow[k0 + WAVE + lane] : 0ull;
                bs_n = in ? brow[k0 + WAVE + lane] : 0u;
            }
            const uint32_t mk_lo = (uint32_t)mk, mk_hi = (uint32_t)(mk >> 32);
#pragma unroll
            for (int kk = 0; kk < WAVE; kk += RU) {
                if (k0 + kk >= np_pad) break;                            // wave-uniform
                double v[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readlane((int)mk_lo, kk + u);
                    const uint32_t m_hi = (uint32_t)__builtin_amdgcn_readlane((int)mk_hi, kk + u);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bs, kk + u);
                    v[u] = band_mask_bit(m_lo, m_hi, lane) ? partial[b + band_mask_rank(m_lo, m_hi, lane)] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) s += v[u];              // ascending pieces (absent ones add +0.0)
            }
        }
        // the heads: lane = record, its carries added in slot order; the totals are handed to their rows one after the other in
        // record order (sorted by row, then slot) — a fixed order, like everything else in this SpMV
        for (uint32_t i0 = sb; i0 < se; i0 += WAVE) {                    // wave-uniform
            if (i0 != sb) {                                              // (more than 64 records in one row block: rare)
                h_jl = 0xFFFFFFFFu;
                h_n = 0;
                h_c = 0.0;
                if (i0 + lane < se) {
                    const RSpill rec = ra.rsp[i0 + lane];
                    h_jl = rec.j & (WAVE - 1);
                    h_first = rec.first;
                    h_n = rec.n;
                    h_c = ra.carry[h_first];
                }
            }
            for (uint32_t k = 1; k < h_n; ++k) h_c += ra.carry[h_first + k];
            const uint32_t cnt = se - i0 < (uint32_t)WAVE ? se - i0 : (uint32_t)WAVE;
            for (uint32_t t = 0; t < cnt; ++t) {                         // wave-uniform
                const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)h_jl, (int)t);
                const int lo = __builtin_amdgcn_readlane(__double2loint(h_c), (int)t);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(h_c), (int)t);
                if (lane == tj) s += __hiloint2double(hi, lo);
            }
        }
        if (j < n_long) {
            if constexpr (ACC) y[r] = y[r] + s;
            else y[r] = s;
        }
    }
}

template <bool ACC>
__global__ __launch_bounds__(256) void band_reduce_kernel(ReduceArgs ra) {
    band_reduce_body<ACC>(ra, blockIdx.x, gridDim.x);
}

// SMALL PLANS (a few tiles per wave: R-MAT 1M, a rank's block of an 8-way cut): every launch of such an SpMV is a few
// microseconds of work behind a latency floor of about five, so the launches that do not depend on each other share one.
// The reduction of the long rows (needs the hot slices and the cold pieces) and the short rows (need only xp; they write their
// own rows of y) are the two halves of ONE grid: blocks [0, reduce_blocks) reduce, the rest walk the short piece — both are
// 256-thread workgroups of independent waves within 64 registers.  Round 5 ran them one after the other (short rows 13.8 us,
// reduction 9.2 us).
template <bool ACC>
__global__ __launch_bounds__(CNT, 8) void band_tail_kernel(ReduceArgs ra, uint32_t reduce_blocks, ColdArgs ca) {
    __shared__ __attribute__((aligned(16))) double stage_s[CNT / WAVE][STG];
    if (blockIdx.x < reduce_blocks) band_reduce_body<ACC>(ra, blockIdx.x, reduce_blocks);
    else band_cold_body<ACC, true>(ca, blockIdx.x - reduce_blocks, stage_s);
}

}  // namespace
}  // namespace sprs_hip
