// Hardware check of sprs_amd/csrc/lanes.hpp: every DPP / permlane-swap exchange against the __shfl form, the group scans
// against a loop, the group sort against std::sort — on random 32- and 64-bit keys, all 64 lanes, several waves.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/probes/lane_ops.out scripts/probes/lane_ops.hip && scripts/probes/lane_ops.out
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../sprs_amd/csrc/lanes.hpp"

using namespace sprs_hip;

// out[0..] per lane: xor 1, 2, 4, 8, 16, 32 (u32), the same for u64 (6 more), below, above, scan16, scan32, scan64
__global__ void exchanges(const uint64_t *in, uint64_t *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t v64 = in[t];
    const uint32_t v = (uint32_t)v64;
    uint64_t *o = out + (uint64_t)t * 24;
    o[0] = lane_xor<1>(v); o[1] = lane_xor<2>(v); o[2] = lane_xor<4>(v); o[3] = lane_xor<8>(v); o[4] = lane_xor<16>(v); o[5] = lane_xor<32>(v);
    o[6] = lane_xor<1>(v64); o[7] = lane_xor<2>(v64); o[8] = lane_xor<4>(v64); o[9] = lane_xor<8>(v64); o[10] = lane_xor<16>(v64); o[11] = lane_xor<32>(v64);
    o[12] = lane_below(v); o[13] = lane_above(v);
    o[14] = group_incl_scan_u32<16>(v & 0xFFFFu); o[15] = group_incl_scan_u32<32>(v & 0xFFFFu); o[16] = group_incl_scan_u32<64>(v & 0xFFFFu);
    o[17] = group_sort<16>(v); o[18] = group_sort<32>(v); o[19] = group_sort<64>(v);
    o[20] = group_sort<16>(v64); o[21] = group_sort<32>(v64); o[22] = group_sort<64>(v64);
    o[23] = 0;
}

int main(int argc, char **argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4096, n = waves * 64;      // (a multiple of 4: four waves per block)
    std::mt19937_64 rng(12345);
    std::vector<uint64_t> in(n), out((size_t)n * 24);
    for (auto &x : in) x = rng();
    for (int i = 0; i < 64; ++i) in[i] = (uint64_t)(i * 3 + 1) | ((uint64_t)(1000 - i) << 32);     // a readable first wave
    uint64_t *d_in, *d_out;
    (void)hipMalloc((void **)&d_in, n * 8);
    (void)hipMalloc((void **)&d_out, (size_t)n * 24 * 8);
    (void)hipMemcpy(d_in, in.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exchanges, dim3(waves / 4), dim3(256), 0, 0, d_in, d_out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("{\"probe\": \"lane_ops\", \"error\": \"launch\"}\n"); return 1; }
    (void)hipMemcpy(out.data(), d_out, (size_t)n * 24 * 8, hipMemcpyDeviceToHost);
    long bad[24] = {0};
    for (int w = 0; w < waves; ++w) {
        const uint64_t *x = &in[(size_t)w * 64];
        auto O = [&](int lane, int k) { return out[((size_t)w * 64 + lane) * 24 + k]; };
        for (int l = 0; l < 64; ++l) {
            for (int b = 0; b < 6; ++b) {
                if (O(l, b) != (uint32_t)x[l ^ (1 << b)]) ++bad[b];
                if (O(l, 6 + b) != x[l ^ (1 << b)]) ++bad[6 + b];
            }
            if (O(l, 12) != (l ? (uint32_t)x[l - 1] : 0u)) ++bad[12];
            if (O(l, 13) != (l < 63 ? (uint32_t)x[l + 1] : 0u)) ++bad[13];
        }
        const int gs[3] = {16, 32, 64};
        for (int gi = 0; gi < 3; ++gi) {
            const int G = gs[gi];
            for (int g0 = 0; g0 < 64; g0 += G) {
                uint32_t s = 0;
                std::vector<uint32_t> k32(G);
                std::vector<uint64_t> k64(G);
                for (int l = 0; l < G; ++l) {
                    s += (uint32_t)x[g0 + l] & 0xFFFFu;
                    if (O(g0 + l, 14 + gi) != s) ++bad[14 + gi];
                    k32[l] = (uint32_t)x[g0 + l];
                    k64[l] = x[g0 + l];
                }
                std::sort(k32.begin(), k32.end());
                std::sort(k64.begin(), k64.end());
                for (int l = 0; l < G; ++l) {
                    if (O(g0 + l, 17 + gi) != k32[l]) ++bad[17 + gi];
                    if (O(g0 + l, 20 + gi) != k64[l]) ++bad[20 + gi];
                }
            }
        }
    }
    const char *names[23] = {"xor1_u32", "xor2_u32", "xor4_u32", "xor8_u32", "xor16_u32", "xor32_u32", "xor1_u64", "xor2_u64", "xor4_u64", "xor8_u64",
                             "xor16_u64", "xor32_u64", "below", "above", "scan16", "scan32", "scan64", "sort16_u32", "sort32_u32", "sort64_u32",
                             "sort16_u64", "sort32_u64", "sort64_u64"};
    long total = 0;
    printf("{\"probe\": \"lane_ops\", \"waves\": %d, \"mismatches\": {", waves);
    for (int k = 0; k < 23; ++k) {
        printf("%s\"%s\": %ld", k ? ", " : "", names[k], bad[k]);
        total += bad[k];
    }
    printf("}, \"ok\": %s}\n", total ? "false" : "true");
    return total ? 1 : 0;
}
