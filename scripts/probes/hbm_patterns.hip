// Hardware probe (developer tool, not part of the library): what the MI355X memory system delivers for the access patterns
// that bound the SpMV and the SpMM — a streaming read, a copy, and random gathers of whole rows of 64 / 128 / 256 bytes from
// tables that fit an L2 (4 MiB), the Infinity Cache (128 MiB) or neither (1.28 GB and up).  The row of a gather is
// hash(i) % rows computed in registers, so the only traffic is the gathered rows; 16 rows in flight per lane like the SpMM
// kernel.  usage: hbm_patterns.out      (prints one JSON line per pattern: best of 5 runs)
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

struct alignas(16) V2 {
    double x, y;
};

__global__ __launch_bounds__(256) void read_kernel(const V2 *__restrict__ src, size_t n, double *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0.0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const V2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < n; i += stride) s += src[i].x + src[i].y;
    if (s == 1.2345e-300) out[0] = s;                   // (keeps the loads)
}

__global__ __launch_bounds__(256) void copy_kernel(const V2 *__restrict__ src, V2 *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__device__ __forceinline__ uint32_t mix(uint32_t v) {
    v ^= v >> 16;
    v *= 0x7feb352du;
    v ^= v >> 15;
    v *= 0x846ca68bu;
    v ^= v >> 16;
    return v;
}

// groups of KP lanes gather rows of KP doubles: gather number q of group gi reads row mix(q) % rows
template <int KP>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ table, uint32_t rows, uint32_t per_group,
                                                     double *__restrict__ out) {
    const uint32_t j = threadIdx.x % KP;
    const uint32_t gi = (blockIdx.x * 256u + threadIdx.x) / KP;
    double s = 0.0;
    for (uint32_t b = 0; b < per_group; b += 16) {
        double x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t r = mix(gi * per_group + b + (uint32_t)u) % rows;
            x[u] = table[(size_t)r * KP + j];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += x[u];
    }
    if (s == 1.2345e-300) out[0] = s;
}

template <typename F>
static double best_ms(F launch) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(a, 0));
        launch();
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;                // (the first run warms up)
    }
    return best;
}

template <int KP>
static void gather_case(const double *table, size_t table_bytes, uint32_t rows, double *out) {
    const uint32_t gathers = 1u << 28;                  // 268 M rows gathered
    const uint32_t groups = 256 * 28 * (256 / KP) * 4, per_group = gathers / groups / 16 * 16;
    const double ms = best_ms([&] {
        hipLaunchKernelGGL(gather_kernel<KP>, dim3(groups / (256 / KP)), dim3(256), 0, 0, table, rows, per_group, out);
    });
    const double n = (double)groups * per_group;
    printf("{\"pattern\": \"gather\", \"row_bytes\": %d, \"table_MB\": %.1f, \"gathers\": %.0f, \"ms\": %.3f, \"Ggathers_per_s\": %.1f, "
           "\"useful_TBs\": %.2f, \"line_TBs_128B\": %.2f}\n",
           KP * 8, (double)rows * KP * 8 / 1e6, n, ms, n / ms / 1e6, n * KP * 8 / ms / 1e9, n * (KP * 8 < 128 ? 128 : KP * 8) / ms / 1e9);
    (void)table_bytes;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    V2 *src, *dst;
    double *out;
    CHECK(hipMalloc((void **)&src, bytes));
    CHECK(hipMalloc((void **)&dst, bytes));
    CHECK(hipMalloc((void **)&out, 64));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMemset(dst, 0, bytes));
    const size_t n = bytes / sizeof(V2);
    for (int blocks : {256 * 8, 256 * 16, 256 * 32}) {
        double ms = best_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, src, n, out); });
        printf("{\"pattern\": \"read\", \"GB\": %.2f, \"workgroups\": %d, \"ms\": %.3f, \"TBs\": %.2f}\n", bytes / 1e9, blocks, ms, bytes / ms / 1e9);
        ms = best_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, n / 2); });
        printf("{\"pattern\": \"copy\", \"GB_read_plus_written\": %.2f, \"workgroups\": %d, \"ms\": %.3f, \"TBs\": %.2f}\n", bytes / 1e9, blocks, ms,
               bytes / ms / 1e9);
    }
    const double *table = (const double *)src;
    for (uint32_t rows : {32768u, 1000000u, 10000000u, 30000000u}) {
        gather_case<8>(table, bytes, rows, out);
        gather_case<16>(table, bytes, rows, out);
        if ((size_t)rows * 256 <= bytes) gather_case<32>(table, bytes, rows, out);
    }
    CHECK(hipFree(src));
    CHECK(hipFree(dst));
    CHECK(hipFree(out));
    return 0;
}
