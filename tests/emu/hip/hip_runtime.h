// TEST INFRASTRUCTURE — not part of the product.
//
// A minimal stand-in for <hip/hip_runtime.h> that lets the kernels of sprs_amd/csrc/*.hip be
// compiled by the HOST clang++ and executed on the CPU, one fiber per GPU thread, so that the
// LOGIC of a kernel (indexing, barriers, wave collectives, plan building) can be debugged in the
// build container, which has no GPU.  It says nothing about speed and little about the memory
// model; every number and every parity claim comes from the real gfx950 build on an MI355X.
//
// Model: a block's threads are ucontext fibers run by one OS thread; a thread runs until it
// reaches a workgroup barrier, a wave collective (__shfl*, __ballot, readlane, wave_barrier) or
// its end.  Collectives are resolved among the lanes of a wave that wait at the same call site
// (lowest code address first, which follows structured control flow), so divergent branches
// behave like the exec-masked hardware.  "Device" memory is host memory behind a guard page
// (reads and writes past the end of an allocation fault at once) and is filled with 0xFF bytes.
#pragma once
#define SPRS_HIP_EMU 1

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// ---- host API -------------------------------------------------------------------------------
enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorInsufficientDriver = 35,
    hipErrorNoDevice = 100,
    hipErrorInvalidDevice = 101,
    hipErrorUnknown = 999
};
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipemuStream;
typedef hipemuStream *hipStream_t;

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern "C++" {
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s = nullptr);
hipError_t hipMemset(void *dst, int v, size_t n);
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s = nullptr);
typedef void *hipDeviceptr_t;
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t s = nullptr);
hipError_t hipMemset2DAsync(void *dst, size_t pitch, int v, size_t width, size_t height, hipStream_t s = nullptr);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }   // a small "chip": several hot workgroups
struct hipemuEvent;
typedef hipemuEvent *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
}

// ---- device side ------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

namespace hipemu {
struct Idx {
    uint32_t x, y, z;
};
struct Fiber;
extern Fiber *cur;
extern Idx g_blockIdx, g_blockDim, g_gridDim;
const Idx &tid_of_current();

enum Op { OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_BARRIER, OP_DPP, OP_PERMLANE16_SWAP, OP_PERMLANE32_SWAP };
// noinline: the return address identifies the call site inside the kernel
// convergent + noduplicate: the host compiler must not clone a call into both arms of a branch (jump threading
// would split the lanes of one logical collective over two call sites)
uint64_t collective(Op op, uint64_t val, int arg, int width) __attribute__((noinline, convergent, noduplicate));
// ... with a second operand per lane (the `old` value of a DPP move, the second register of a permlane swap)
uint64_t collective2(Op op, uint64_t val, uint64_t val2, int arg, int width) __attribute__((noinline, convergent, noduplicate));
void syncthreads() __attribute__((noinline, convergent, noduplicate));
// a wave spinning on an LDS word written by another wave of the block (spgemm.hip, token hand-over): the fiber steps
// aside and is resumed on the scheduler's next sweep over the block
void poll_yield() __attribute__((noinline));

struct Launcher {
    virtual void call() = 0;
    virtual ~Launcher() {}
};
void launch(const char *name, dim3 grid, dim3 block, Launcher &l);

template <typename F>
struct LambdaLauncher : Launcher {
    F f;
    explicit LambdaLauncher(F f_) : f(f_) {}
    void call() override { f(); }
};
template <typename F>
inline void launch_lambda(const char *name, dim3 grid, dim3 block, F f) {
    LambdaLauncher<F> l(f);
    launch(name, grid, block, l);
}

template <typename T>
inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8 && std::is_trivially_copyable<T>::value, "collective payload");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <typename T>
inline T from_bits(uint64_t b) {
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}
}  // namespace hipemu

#define threadIdx (hipemu::tid_of_current())
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipemu::launch_lambda(#kern, dim3(grid), dim3(block), [&]() { (kern)(__VA_ARGS__); })

#define HIPEMU_INLINE inline __attribute__((always_inline, convergent))
HIPEMU_INLINE void __syncthreads() { hipemu::syncthreads(); }
HIPEMU_INLINE unsigned long long __ballot(int pred) { return hipemu::collective(hipemu::OP_BALLOT, pred ? 1 : 0, 0, 64); }
template <typename T>
HIPEMU_INLINE T __shfl(T v, int src, int width = 64) {
    return hipemu::from_bits<T>(hipemu::collective(hipemu::OP_SHFL, hipemu::to_bits(v), src, width));
}
template <typename T>
HIPEMU_INLINE T __shfl_up(T v, unsigned delta, int width = 64) {
    return hipemu::from_bits<T>(hipemu::collective(hipemu::OP_SHFL_UP, hipemu::to_bits(v), (int)delta, width));
}
template <typename T>
HIPEMU_INLINE T __shfl_down(T v, unsigned delta, int width = 64) {
    return hipemu::from_bits<T>(hipemu::collective(hipemu::OP_SHFL_DOWN, hipemu::to_bits(v), (int)delta, width));
}
template <typename T>
HIPEMU_INLINE T __shfl_xor(T v, int mask, int width = 64) {
    return hipemu::from_bits<T>(hipemu::collective(hipemu::OP_SHFL_XOR, hipemu::to_bits(v), mask, width));
}
// The gfx9 DPP move (v_mov_b32_dpp) as the kernels use it through __builtin_amdgcn_update_dpp: the lane exchanges of the shipped
// sources run here as written — quad_perm, row_shl / row_shr / row_ror, wave_shl:1 / wave_shr:1, row_mirror / row_half_mirror,
// row_bcast:15 / 31, row and bank masks, bound_ctrl (hipemu.cpp: resolve_one) — and so do v_permlane16_swap / v_permlane32_swap
// of gfx950.  scripts/probes/lane_ops.hip holds the same sequences against __shfl on the hardware.
HIPEMU_INLINE int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int arg = (ctrl & 0x1ff) | ((row_mask & 0xf) << 12) | ((bank_mask & 0xf) << 16) | (bound_ctrl ? 1 << 20 : 0);
    return (int)(uint32_t)hipemu::collective2(hipemu::OP_DPP, (uint32_t)src, (uint32_t)old, arg, 64);
}
struct hipemu_uint2 {
    unsigned v[2];
    unsigned operator[](int i) const { return v[i]; }
};
HIPEMU_INLINE hipemu_uint2 __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
    const uint64_t r = hipemu::collective2(hipemu::OP_PERMLANE16_SWAP, a, b, 0, 64);
    return hipemu_uint2{{(unsigned)r, (unsigned)(r >> 32)}};
}
HIPEMU_INLINE hipemu_uint2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const uint64_t r = hipemu::collective2(hipemu::OP_PERMLANE32_SWAP, a, b, 0, 64);
    return hipemu_uint2{{(unsigned)r, (unsigned)(r >> 32)}};
}
// v_mbcnt_lo / v_mbcnt_hi: mask bits below the lane (plus the addend); inverse ballot: the lane's bit of a wave-uniform mask
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned add) {
    const unsigned lane = hipemu::tid_of_current().x & 63u;
    return add + (unsigned)__builtin_popcount(lane >= 32 ? m : m & ((1u << lane) - 1u));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned add) {
    const unsigned lane = hipemu::tid_of_current().x & 63u;
    return add + (lane <= 32 ? 0u : (unsigned)__builtin_popcount(m & ((1u << (lane - 32)) - 1u)));
}
inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> (hipemu::tid_of_current().x & 63u)) & 1ull; }
HIPEMU_INLINE int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane, 64); }
HIPEMU_INLINE int __builtin_amdgcn_readfirstlane(int v) {
    const unsigned long long m = __ballot(1);
    return __shfl(v, __builtin_ctzll(m), 64);
}
HIPEMU_INLINE void __builtin_amdgcn_wave_barrier() { (void)hipemu::collective(hipemu::OP_BARRIER, 0, 0, 64); }
HIPEMU_INLINE void __builtin_amdgcn_s_barrier() { hipemu::syncthreads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// the kernels' hand-written waits / LDS-only barriers (scan.hpp, spgemm.hip) map onto these
#define SPRS_LDS_BARRIER() hipemu::syncthreads()
#define SPRS_WAIT_ALL() ((void)0)
#define SPRS_LDS_FENCE() ((void)0)
#define SPRS_POLL_PAUSE() hipemu::poll_yield()

#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
// agent-scope 8-byte hand-offs (gauss_seidel.hip): fibers switch cooperatively, a plain access is atomic
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
inline long long __double_as_longlong(double v) { long long b; memcpy(&b, &v, 8); return b; }
inline double __longlong_as_double(long long b) { double v; memcpy(&v, &b, 8); return v; }
inline int __double2loint(double v) { long long b; memcpy(&b, &v, 8); return (int)(unsigned)(b & 0xFFFFFFFFll); }
inline int __double2hiint(double v) { long long b; memcpy(&b, &v, 8); return (int)(unsigned)((unsigned long long)b >> 32); }
inline double __hiloint2double(int hi, int lo) { long long b = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); double v; memcpy(&v, &b, 8); return v; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline long long clock64() { return 0; }
inline long long wall_clock64() { return 0; }

// atomics: fibers are switched cooperatively, so plain read-modify-write is atomic
template <typename T, typename U>
inline T atomicAdd(T *p, U v) {
    const T o = *p;
    *p = (T)(o + (T)v);
    return o;
}
template <typename T, typename U>
inline T atomicSub(T *p, U v) {
    const T o = *p;
    *p = (T)(o - (T)v);
    return o;
}
template <typename T, typename U>
inline T atomicMin(T *p, U v) {
    const T o = *p;
    if ((T)v < o) *p = (T)v;
    return o;
}
template <typename T, typename U>
inline T atomicMax(T *p, U v) {
    const T o = *p;
    if ((T)v > o) *p = (T)v;
    return o;
}
template <typename T, typename U>
inline T atomicOr(T *p, U v) {
    const T o = *p;
    *p = (T)(o | (T)v);
    return o;
}
template <typename T, typename U>
inline T atomicAnd(T *p, U v) {
    const T o = *p;
    *p = (T)(o & (T)v);
    return o;
}
template <typename T, typename U>
inline T atomicExch(T *p, U v) {
    const T o = *p;
    *p = (T)v;
    return o;
}
template <typename T, typename U, typename V>
inline T atomicCAS(T *p, U cmp, V v) {
    const T o = *p;
    if (o == (T)cmp) *p = (T)v;
    return o;
}
