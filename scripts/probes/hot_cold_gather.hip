// Hardware probe (developer tool, not part of the library): can an L2 keep the HOT rows of an SpMM's right-hand side while the
// cold rows stream past it?  Lane groups of 16 gather 128-byte rows (k = 16 doubles) like spmm_stream_kernel, 16 in flight per
// lane.  A share HOT_PCT of the gathers goes to a hot table of `hot_rows` rows (32 K rows = 4 MB: what R-MAT 10M's 32 K most
// popular columns hold, 45 % of its entries), the rest to a 1.28 GB cold table.  Policies for the COLD gathers:
//   0 plain loads (what the library does)     1 non-temporal loads     2 sc1 (system-coherent, L1-bypassing) loads
// Every gather is issued as TWO unconditional loads — the hot one and the cold one, the lane's unused one aimed at row 0 of the hot
// table (an L1 hit) — because a load under a per-lane condition is compiled as a branch with a full wait.
// One JSON line per (policy, hot rows, hot share): G gathers/s; compare with the all-cold and all-hot rates printed first.
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

__device__ __forceinline__ uint32_t mix(uint32_t v) {
    v ^= v >> 16;
    v *= 0x7feb352du;
    v ^= v >> 15;
    v *= 0x846ca68bu;
    v ^= v >> 16;
    return v;
}

template <int POLICY>
__device__ __forceinline__ double cold_load(const double *p) {
    if (POLICY == 1) return __builtin_nontemporal_load(p);
    if (POLICY == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *p;
}

constexpr int KP = 16;

template <int POLICY>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ hot, uint32_t hot_rows, const double *__restrict__ cold,
                                                     uint32_t cold_rows, uint32_t hot_pct, uint32_t per_group, double *__restrict__ out) {
    const uint32_t j = threadIdx.x % KP;
    const uint32_t gi = (blockIdx.x * 256u + threadIdx.x) / KP;
    double s = 0.0;
    for (uint32_t b = 0; b < per_group; b += 16) {
        double xh[16], xc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t q = gi * per_group + b + (uint32_t)u;
            const uint32_t h = mix(q);
            const bool is_hot = mix(q ^ 0x9e3779b9u) % 100u < hot_pct;
            const uint32_t rh = is_hot ? h % hot_rows : 0u;
            const uint32_t rc = is_hot ? 0u : h % cold_rows;
            xh[u] = hot[(size_t)rh * KP + j];
            xc[u] = is_hot ? 0.0 : 1.0;                       // (keeps the selection alive)
            xc[u] *= cold_load<POLICY>(is_hot ? hot + j : cold + (size_t)rc * KP + j);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += xh[u] + xc[u];
    }
    if (s == 1.2345e-300) out[0] = s;
}

template <int POLICY>
static void run(const double *hot, uint32_t hot_rows, const double *cold, uint32_t cold_rows, uint32_t hot_pct, double *out) {
    const uint32_t gathers = 1u << 27;
    const uint32_t groups = 256 * 28 * (256 / KP) * 4, per_group = gathers / groups / 16 * 16;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(gather_kernel<POLICY>, dim3(groups / (256 / KP)), dim3(256), 0, 0, hot, hot_rows, cold, cold_rows, hot_pct, per_group, out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double n = (double)groups * per_group;
    printf("{\"cold_policy\": %d, \"hot_rows\": %u, \"hot_MB\": %.1f, \"hot_pct\": %u, \"gathers\": %.0f, \"ms\": %.3f, \"Ggathers_per_s\": %.1f}\n", POLICY,
           hot_rows, hot_rows * 128.0 / 1e6, hot_pct, n, best, n / best / 1e6);
    fflush(stdout);
}

int main() {
    const uint32_t cold_rows = 10000000u;
    double *cold, *hot, *out;
    CHECK(hipMalloc((void **)&cold, (size_t)cold_rows * 128));
    CHECK(hipMalloc((void **)&hot, (size_t)(1u << 20) * 128));
    CHECK(hipMalloc((void **)&out, 64));
    CHECK(hipMemset(cold, 1, (size_t)cold_rows * 128));
    CHECK(hipMemset(hot, 1, (size_t)(1u << 20) * 128));
    run<0>(hot, 32768, cold, cold_rows, 0, out);         // all cold
    run<0>(hot, 32768, cold, cold_rows, 100, out);       // all hot, 4 MB
    for (uint32_t hr : {8192u, 16384u, 32768u, 65536u})
        for (uint32_t pct : {30u, 45u, 60u}) {
            run<0>(hot, hr, cold, cold_rows, pct, out);
            run<1>(hot, hr, cold, cold_rows, pct, out);
            run<2>(hot, hr, cold, cold_rows, pct, out);
        }
    return 0;
}
