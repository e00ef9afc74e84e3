// Hardware probe (developer tool, not part of the library): what the partial-sum stores of the banded SpMV's hot kernel cost
// beside its stream, and what more loads in flight per wave would buy.  The kernel has the hot kernel's shape — one
// 1024-thread workgroup per CU (144 KiB of dynamic LDS keeps a second one out), 16 waves, every wave walks RANGES of 4
// consecutive tiles of 5 120 bytes (four 16-byte value loads + one 16-byte id load per lane, non-temporal), ranges dealt
// round-robin to the waves, DEPTH tiles requested ahead — and after every tile it writes C consecutive doubles at the running
// offset tile * C of an output array (the sums of the rows that end in the tile), in one of several ways:
//   0 no stores              1 8-byte non-temporal stores, lane = sum (what the library does)
//   2 8-byte plain stores    3 16-byte non-temporal stores, lane = two sums (offset rounded down to even)
//   4 whole 128-byte lines only: the sums are parked and written when 16-aligned groups are complete (8-byte nt)
//   5 as 4 with 16-byte stores   6 8-byte sc1 (write-through) stores
// ring_MB: the output offsets wrap around a window of that many MB (0: every sum has its own address).
//   9 as 1, but the sums of a whole range of 4 tiles parked in LDS and written in one burst at its end (C <= 128)
//   7 as 1 and 8 as 6, but as exactly two predicated store instructions per tile (C <= 128): counted by the compiler's waits
// ORDER 0: the stores of a step in front of its request (default), 1: behind it.
// usage: stream_store.out [GiB of stream, default 3]     one JSON line per (mode, order, depth, C): best of 4 runs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);   \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64, THREADS = 1024, WAVES = THREADS / WAVE, RUN = 4;

struct Tile {
    dbl2 v[4];
    u32x4 c;
};

__device__ __forceinline__ void request(Tile &t, const double *vals, const uint32_t *ids, uint64_t w, uint32_t lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) t.v[p] = __builtin_nontemporal_load((const dbl2 *)(vals + w * 512 + p * 128 + lane * 2));
    t.c = __builtin_nontemporal_load((const u32x4 *)(ids + w * 256 + lane * 4));
}

__device__ __forceinline__ double fold(const Tile &t) {
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) s += t.v[p][0] * (double)(t.c[p] & 0xFFFFu) + t.v[p][1] * (double)(t.c[p] >> 16);
    return s;
}

template <int MODE, int DEPTH, int ORDER>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 5))) void probe(const double *__restrict__ vals, const uint32_t *__restrict__ ids,
                                                                                         uint64_t ntiles, uint32_t C, double *__restrict__ out, uint64_t ring) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x / 64u;
    double *stage = lds + wave * 256;                                    // 2 KiB per wave
    // the workgroup's share of the tiles, cut into ranges of RUN tiles dealt round-robin to the waves
    const uint64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint64_t t0 = per * blockIdx.x, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    const uint64_t n = t1 > t0 ? t1 - t0 : 0;
    auto tile_of = [&](uint64_t i) -> uint64_t {                         // i-th tile of this wave
        const uint64_t r = i / RUN, q = i % RUN;
        return (r * WAVES + wave) * RUN + q;
    };
    Tile tl[DEPTH];
    uint64_t i = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) request(tl[d], vals, ids, tile_of(d) < n ? t0 + tile_of(d) : 0, lane);
    uint32_t parked = 0;                                                 // modes 4, 5: sums waiting in the stage
    uint64_t parked_at = 0;
    bool done = false;
    for (; !done; i += DEPTH) {
        // the DEPTH register sets take turns (no copies: a copy of a tile in flight would wait for it)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint64_t t = tile_of(i + d);
            if (t >= n) {
                done = true;
                break;
            }
            // order of the memory operations of one step: the stores first, then the request DEPTH tiles ahead — loads and
            // stores count on ONE in-order vmcnt on gfx950, so a wait for the next tile's loads also waits for every store
            // issued before them; issued in front of the newest request the stores have a whole step to complete (ORDER 1: the
            // request first, which makes the next wait drain the stores just issued)
            const double s = fold(tl[d]);
            auto advance = [&]() {
                // (always issued — past the end the wave re-reads its last tile: behind a condition the compiler cannot count the
                // operations in flight and every wait becomes vmcnt(0))
                const uint64_t tn = tile_of(i + d + DEPTH);
                request(tl[d], vals, ids, t0 + (tn < n ? tn : t), lane);
            };
            if (ORDER == 1) advance();
            const uint64_t o = ((t0 + t) * C) % ring;         // ring: doubles of the output window that is written over and over
            if (MODE == 1 || MODE == 2 || MODE == 6) {
                for (uint32_t j = lane; j < C; j += WAVE) {
                    if (MODE == 1) __builtin_nontemporal_store(s + j, out + o + j);
                    else if (MODE == 2) out[o + j] = s + j;
                    else __hip_atomic_store(out + o + j, s + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (MODE == 9) {
                // the sums of a whole RANGE (RUN tiles) parked in LDS and written in ONE burst when the range ends: the same bytes in
                // RUN times fewer, longer bursts
                const bool first = ((i + d) % RUN) == 0, last = ((i + d) % RUN) == RUN - 1;
                double *big = lds + wave * 512;
                if (first) {
                    parked = 0;
                    parked_at = o;
                }
                for (uint32_t j = lane; j < C; j += WAVE) big[(parked + j) & 511u] = s + j;
                parked += C;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (last) {
                    for (uint32_t j = lane; j < parked; j += WAVE) __builtin_nontemporal_store(big[j & 511u], out + parked_at + j);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            } else if (MODE == 7 || MODE == 8) {
                // exactly two store instructions per tile, no loop: the compiler can count them, and a wait for the loads of
                // the next tile leaves the stores issued after those loads in flight
                const uint32_t j0 = lane, j1 = lane + WAVE;
                if (MODE == 7) {
                    if (j0 < C) __builtin_nontemporal_store(s + j0, out + o + j0);
                    if (j1 < C) __builtin_nontemporal_store(s + j1, out + o + j1);
                } else {
                    if (j0 < C) __hip_atomic_store(out + o + j0, s + j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (j1 < C) __hip_atomic_store(out + o + j1, s + j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (MODE == 3) {
                const uint64_t oe = o & ~1ull;
                for (uint32_t j = lane * 2; j < C; j += WAVE * 2) __builtin_nontemporal_store(dbl2{s + j, s}, (dbl2 *)(out + oe + j));
            } else if (MODE == 4 || MODE == 5) {
                // park the tile's sums behind what is parked already (consecutive tiles of a range write consecutive sums)
                const bool first = ((i + d) % RUN) == 0;
                if (first) {
                    parked = 0;
                    parked_at = o;
                }
                for (uint32_t j = lane; j < C; j += WAVE) stage[(parked + j) & 255u] = s + j;
                parked += C;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool last = ((i + d) % RUN) == RUN - 1;
                // complete lines: from parked_at up to the last multiple of 16 below parked_at + parked (everything when the range ends)
                uint64_t end = last ? parked_at + parked : ((parked_at + parked) & ~15ull);
                if (end > parked_at) {
                    const uint32_t cnt = (uint32_t)(end - parked_at);
                    if (MODE == 4) {
                        for (uint32_t j = lane; j < cnt; j += WAVE) __builtin_nontemporal_store(stage[j & 255u], out + parked_at + j);
                    } else {
                        const uint64_t pe = parked_at & ~1ull;
                        for (uint32_t j = lane * 2; j < cnt; j += WAVE * 2)
                            __builtin_nontemporal_store(dbl2{stage[j & 255u], stage[(j + 1) & 255u]}, (dbl2 *)(out + pe + j));
                    }
                    // what stays parked moves to the front (fewer than 16 sums)
                    const uint32_t left = parked - cnt;
                    double keep = 0.0;
                    if (lane < left) keep = stage[(cnt + lane) & 255u];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < left) stage[lane] = keep;
                    parked = left;
                    parked_at = end;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            } else if (s == 1.2345e-300) {
                out[0] = s;
            }
            if (ORDER == 0) advance();
        }
    }
}

template <int MODE, int DEPTH, int ORDER = 0>
static void run(const double *vals, const uint32_t *ids, uint64_t ntiles, uint32_t C, double *out, int ncu, int rounds, uint64_t ring_mb = 0) {
    const uint64_t ring = ring_mb ? ring_mb * 131072ull : ~0ull >> 8;
    const int lds = 144 * 1024;
    CHECK(hipFuncSetAttribute((const void *)probe<MODE, DEPTH, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((probe<MODE, DEPTH, ORDER>), dim3(ncu * rounds), dim3(THREADS), lds, 0, vals, ids, ntiles, C, out, ring);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double rd = (double)ntiles * 5120.0, wr = MODE ? (double)ntiles * C * 8.0 : 0.0;
    printf("{\"mode\": %d, \"order\": %d, \"depth\": %d, \"sums_per_tile\": %u, \"rounds\": %d, \"ring_MB\": %llu, \"ms\": %.4f, \"read_GB\": %.3f, \"write_GB\": %.3f, \"read_TBs\": %.3f, \"total_TBs\": %.3f}\n",
           MODE, ORDER, DEPTH, C, rounds, (unsigned long long)ring_mb, best, rd / 1e9, wr / 1e9, rd / best / 1e9, (rd + wr) / best / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 3.0;
    const uint64_t ntiles = (uint64_t)(gib * 1073741824.0 / 5120.0);
    double *vals, *out;
    uint32_t *ids;
    CHECK(hipMalloc((void **)&vals, ntiles * 4096 + 4096));
    CHECK(hipMalloc((void **)&ids, ntiles * 1024 + 4096));
    CHECK(hipMalloc((void **)&out, ntiles * 512 * 8 + 8192));
    CHECK(hipMemset(vals, 0, ntiles * 4096));
    CHECK(hipMemset(ids, 0, ntiles * 1024));
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    for (int rounds : {2}) {
        run<0, 1>(vals, ids, ntiles, 0, out, ncu, rounds);
        run<0, 2>(vals, ids, ntiles, 0, out, ncu, rounds);
        for (uint32_t C : {6u, 24u, 72u}) {
            run<1, 1>(vals, ids, ntiles, C, out, ncu, rounds);
            run<1, 2>(vals, ids, ntiles, C, out, ncu, rounds);
            run<9, 1>(vals, ids, ntiles, C, out, ncu, rounds);
            run<9, 2>(vals, ids, ntiles, C, out, ncu, rounds);
            run<4, 1>(vals, ids, ntiles, C, out, ncu, rounds);
        }
    }
    return 0;
}
