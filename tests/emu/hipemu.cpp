// TEST INFRASTRUCTURE — runtime of the CPU kernel emulator (see hip/hip_runtime.h).
#include "hip/hip_runtime.h"

#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

namespace hipemu {

enum State { READY, AT_BARRIER, AT_COLLECTIVE, AT_POLL, DONE };

struct Fiber {
    void *sp = nullptr;      // saved stack pointer while the fiber is not running
    Idx tid;
    State state = DONE;
    // pending collective
    Op op;
    uint64_t val = 0, val2 = 0, result = 0;
    int arg = 0, width = 64;
    const void *site = nullptr;
};

Fiber *cur = nullptr;
Idx g_blockIdx{0, 0, 0}, g_blockDim{1, 1, 1}, g_gridDim{1, 1, 1};
static Idx g_host_tid{0, 0, 0};

const Idx &tid_of_current() { return cur ? cur->tid : g_host_tid; }

static void *g_main_sp = nullptr;

// Minimal context switch (System V x86-64: callee-saved registers + stack pointer).  glibc's swapcontext
// makes a sigprocmask system call per switch, which dominated the run time of the emulated kernels.
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");
static Launcher *g_launcher = nullptr;
static const char *g_kernel = "(none)";
static std::vector<Fiber> g_fibers;
static std::vector<char *> g_stacks;
static const size_t STACK_BYTES = 256 * 1024;
static std::recursive_mutex g_mu;   // one launch at a time

static void yield_to_scheduler() {
    Fiber *me = cur;
    hipemu_switch(&me->sp, g_main_sp);
}

void syncthreads() {
    cur->state = AT_BARRIER;
    yield_to_scheduler();
}

void poll_yield() {
    cur->state = AT_POLL;
    yield_to_scheduler();
}

uint64_t collective(Op op, uint64_t val, int arg, int width) {
    Fiber *me = cur;
    me->op = op;
    me->val = val;
    me->arg = arg;
    me->width = width;
    me->site = __builtin_return_address(0);
    me->state = AT_COLLECTIVE;
    yield_to_scheduler();
    return me->result;
}

uint64_t collective2(Op op, uint64_t val, uint64_t val2, int arg, int width) {
    Fiber *me = cur;
    me->op = op;
    me->val = val;
    me->val2 = val2;
    me->arg = arg;
    me->width = width;
    me->site = __builtin_return_address(0);
    me->state = AT_COLLECTIVE;
    yield_to_scheduler();
    return me->result;
}

// source lane of a gfx9 DPP control for `lane` (-1: no valid source: outside the row / the wave)
static int dpp_source(int ctrl, int lane) {
    const int row = lane & ~15, i = lane & 15;
    if (ctrl <= 0xff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);                    // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10f) return i + (ctrl & 15) < 16 ? lane + (ctrl & 15) : -1;  // row_shl:n
    if (ctrl >= 0x111 && ctrl <= 0x11f) return i - (ctrl & 15) >= 0 ? lane - (ctrl & 15) : -1;  // row_shr:n
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((i - (ctrl & 15)) & 15);                  // row_ror:n
    if (ctrl == 0x130) return lane + 1 < 64 ? lane + 1 : -1;                                    // wave_shl:1
    if (ctrl == 0x134) return (lane + 1) & 63;                                                  // wave_rol:1
    if (ctrl == 0x138) return lane - 1;                                                         // wave_shr:1
    if (ctrl == 0x13c) return (lane - 1) & 63;                                                  // wave_ror:1
    if (ctrl == 0x140) return row + 15 - i;                                                     // row_mirror
    if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));                                   // row_half_mirror
    if (ctrl == 0x142) return row >= 16 ? row - 1 : -1;                                         // row_bcast:15 (into the next row)
    if (ctrl == 0x143) return lane >= 32 ? 31 : -1;                                             // row_bcast:31 (into rows 2 and 3)
    fprintf(stderr, "hipemu: DPP control 0x%x is not modelled\n", ctrl);
    abort();
}

static void trampoline() {
    g_launcher->call();
    cur->state = DONE;
    hipemu_switch(&cur->sp, g_main_sp);
    abort();   // a finished fiber is never resumed
}

static void resume(Fiber &f) {
    cur = &f;
    hipemu_switch(&g_main_sp, f.sp);
    cur = nullptr;
}

// resolve the pending collectives of the wave [w0, w1): the group at the lowest call site
static bool resolve_one(size_t w0, size_t w1) {
    const void *site = nullptr;
    for (size_t i = w0; i < w1; ++i) {
        Fiber &f = g_fibers[i];
        if (f.state == AT_COLLECTIVE && (!site || f.site < site)) site = f.site;
    }
    if (!site) return false;
    static const bool trace = getenv("HIPEMU_TRACE") != nullptr;
    bool in[64] = {false};
    uint64_t vals[64] = {0}, vals2[64] = {0};
    unsigned long long ballot = 0;
    for (size_t i = w0; i < w1; ++i) {
        Fiber &f = g_fibers[i];
        const size_t l = i - w0;
        in[l] = f.state == AT_COLLECTIVE && f.site == site;
        vals[l] = in[l] ? f.val : 0;
        vals2[l] = in[l] ? f.val2 : 0;
        if (in[l] && f.op == OP_BALLOT && f.val) ballot |= 1ull << l;
    }
    if (trace) {
        unsigned long long part = 0, live = 0;
        for (size_t i = w0; i < w1; ++i) {
            if (in[i - w0]) part |= 1ull << (i - w0);
            if (g_fibers[i].state != DONE) live |= 1ull << (i - w0);
        }
        if (part != live)
            fprintf(stderr, "hipemu: %s block %u wave %zu: collective at %p with lanes %016llx of live %016llx\n", g_kernel,
                    g_blockIdx.x, w0 / 64, site, part, live);
    }
    for (size_t i = w0; i < w1; ++i) {
        if (!in[i - w0]) continue;
        Fiber &f = g_fibers[i];
        const int lane = (int)(i - w0), w = f.width > 0 ? f.width : 64;
        const int seg = lane / w * w;
        int src = lane;
        switch (f.op) {
            case OP_BALLOT: f.result = ballot; break;
            case OP_BARRIER: f.result = 0; break;
            case OP_SHFL: src = seg + (((f.arg % w) + w) % w); break;
            case OP_SHFL_UP: src = lane - f.arg < seg ? lane : lane - f.arg; break;
            case OP_SHFL_DOWN: src = lane + f.arg >= seg + w ? lane : lane + f.arg; break;
            case OP_SHFL_XOR: src = (lane ^ f.arg) >= seg + w || (lane ^ f.arg) < seg ? lane : (lane ^ f.arg); break;
            case OP_DPP: {
                // v_mov_b32_dpp: a lane outside the row / bank masks keeps `old`; one whose source lane does not exist (or is
                // not executing) gets 0 with bound_ctrl and keeps `old` without
                const int ctrl = f.arg & 0x1ff, row_mask = (f.arg >> 12) & 0xf, bank_mask = (f.arg >> 16) & 0xf;
                const bool bound = (f.arg >> 20) & 1;
                const bool written = ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane & 15) >> 2)) & 1);
                const int s = dpp_source(ctrl, lane);
                const bool ok = s >= 0 && (size_t)s < w1 - w0 && in[s];
                f.result = !written ? f.val2 : ok ? vals[s] : bound ? 0 : f.val2;
                break;
            }
            case OP_PERMLANE16_SWAP: {
                // odd rows of the first operand <-> even rows of the second; result = new first | new second << 32
                const bool odd = (lane >> 4) & 1;
                const uint64_t a = odd ? vals2[lane - 16] : f.val, b = odd ? f.val2 : vals[lane + 16];
                f.result = (a & 0xffffffffull) | (b << 32);
                break;
            }
            case OP_PERMLANE32_SWAP: {
                // upper half of the first operand <-> lower half of the second
                const bool up = lane >= 32;
                const uint64_t a = up ? vals2[lane - 32] : f.val, b = up ? f.val2 : vals[lane + 32];
                f.result = (a & 0xffffffffull) | (b << 32);
                break;
            }
        }
        if (f.op == OP_DPP || f.op == OP_PERMLANE16_SWAP || f.op == OP_PERMLANE32_SWAP) {
            f.state = READY;
            continue;
        }
        if (f.op != OP_BALLOT && f.op != OP_BARRIER) {
            // a source lane that is inactive (exited, or elsewhere in a divergent branch) yields the own value
            const bool ok = src >= 0 && (size_t)src < w1 - w0 && in[src];
            f.result = ok ? vals[src] : f.val;
        }
        f.state = READY;
    }
    return true;
}

static void run_block(size_t nthreads) {
    for (size_t i = 0; i < nthreads; ++i) {
        Fiber &f = g_fibers[i];
        // fresh stack: six zeroed callee-saved registers, then the entry point for hipemu_switch's `ret`
        // (leaves rsp = top - 8, the alignment a called function expects), then a null return address
        uint64_t *top = (uint64_t *)(g_stacks[i] + STACK_BYTES);
        top[-1] = 0;
        top[-2] = (uint64_t)(uintptr_t)&trampoline;
        for (int r = 3; r <= 8; ++r) top[-r] = 0;
        f.sp = (void *)(top - 8);
        f.state = READY;
    }
    // HIPEMU_WAVE_ORDER=reverse|rotate: the order in which the waves of a block get their turn (the hardware promises
    // none; kernels that hand work from wave to wave must not depend on it)
    static const char *order_env = getenv("HIPEMU_WAVE_ORDER");
    static const int order = !order_env ? 0 : order_env[0] == 'r' && order_env[1] == 'e' ? 1 : 2;
    const size_t nwaves = (nthreads + 63) / 64;
    size_t mult = 1;                                    // a multiplier coprime with the number of waves: wi -> wave is a bijection
    for (size_t m : {7, 5, 3, 11, 13}) {
        size_t a = m, b = nwaves;
        while (b) { const size_t t = a % b; a = b; b = t; }
        if (a == 1) { mult = m; break; }
    }
    size_t sweep = 0, idle_sweeps = 0;
    for (;;) {
        bool progressed = false;
        for (size_t wi = 0; wi < nwaves; ++wi) {
            const size_t wv = order == 0 ? wi : order == 1 ? nwaves - 1 - wi : (wi * mult + sweep * 3 + 1) % nwaves;
            const size_t w0 = wv * 64;
            const size_t w1 = w0 + 64 < nthreads ? w0 + 64 : nthreads;
            for (;;) {
                for (size_t i = w0; i < w1; ++i)
                    if (g_fibers[i].state == READY) {
                        resume(g_fibers[i]);
                        if (g_fibers[i].state != AT_POLL) progressed = true;
                    }
                if (!resolve_one(w0, w1)) break;
                progressed = true;
            }
        }
        ++sweep;
        size_t done = 0, waiting = 0, polling = 0;
        for (size_t i = 0; i < nthreads; ++i) {
            done += g_fibers[i].state == DONE;
            waiting += g_fibers[i].state == AT_BARRIER;
            polling += g_fibers[i].state == AT_POLL;
        }
        if (done == nthreads) return;
        if (polling) {
            // pollers go again; barrier waiters stay put until nobody polls any more (the pollers have not arrived yet)
            idle_sweeps = progressed ? 0 : idle_sweeps + 1;
            if (idle_sweeps > 4) {
                fprintf(stderr, "hipemu: kernel %s: block %u stuck (%zu threads poll a word nobody writes)\n", g_kernel, g_blockIdx.x, polling);
                abort();
            }
            for (size_t i = 0; i < nthreads; ++i)
                if (g_fibers[i].state == AT_POLL) g_fibers[i].state = READY;
            continue;
        }
        if (!waiting) {
            fprintf(stderr, "hipemu: kernel %s: block stuck (no thread runnable)\n", g_kernel);
            abort();
        }
        for (size_t i = 0; i < nthreads; ++i)
            if (g_fibers[i].state == AT_BARRIER) g_fibers[i].state = READY;
    }
}

void launch(const char *name, dim3 grid, dim3 block, Launcher &l) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    const size_t nthreads = (size_t)block.x * block.y * block.z;
    if (!nthreads || !grid.x || !grid.y || !grid.z) return;
    if (nthreads > 1024) {
        fprintf(stderr, "hipemu: kernel %s launched with %zu threads per block\n", name, nthreads);
        abort();
    }
    if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    while (g_stacks.size() < nthreads) {
        void *s = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (s == MAP_FAILED) abort();
        g_stacks.push_back((char *)s);
    }
    size_t t = 0;
    for (uint32_t z = 0; z < block.z; ++z)
        for (uint32_t y = 0; y < block.y; ++y)
            for (uint32_t x = 0; x < block.x; ++x) g_fibers[t++].tid = Idx{x, y, z};
    g_launcher = &l;
    g_kernel = name;
    g_blockDim = Idx{block.x, block.y, block.z};
    g_gridDim = Idx{grid.x, grid.y, grid.z};
    for (uint32_t z = 0; z < grid.z; ++z)
        for (uint32_t y = 0; y < grid.y; ++y)
            for (uint32_t x = 0; x < grid.x; ++x) {
                g_blockIdx = Idx{x, y, z};
                run_block(nthreads);
            }
    g_launcher = nullptr;
    g_kernel = "(none)";
}

// ---- memory: every allocation ends at a guard page ----------------------------------------------
struct Alloc {
    void *base;
    size_t total;
    size_t bytes;
};
static std::map<void *, Alloc> g_allocs;
static std::mutex g_alloc_mu;

static void on_fault(int sig, siginfo_t *si, void *) {
    char buf[512];
    int n;
    if (cur)
        n = snprintf(buf, sizeof buf,
                     "hipemu: signal %d at address %p inside kernel %s, block (%u,%u,%u), thread (%u,%u,%u)\n", sig,
                     si->si_addr, g_kernel, g_blockIdx.x, g_blockIdx.y, g_blockIdx.z, cur->tid.x, cur->tid.y, cur->tid.z);
    else
        n = snprintf(buf, sizeof buf, "hipemu: signal %d at address %p on the host side\n", sig, si->si_addr);
    if (n > 0) (void)!write(2, buf, (size_t)n);
    signal(sig, SIG_DFL);
    raise(sig);
}

static void install_handler() {
    static bool done = false;
    if (done) return;
    done = true;
    static char altstack[65536];
    stack_t ss;
    ss.ss_sp = altstack;
    ss.ss_size = sizeof altstack;
    ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
}

}  // namespace hipemu

using namespace hipemu;

hipError_t hipMalloc(void **p, size_t bytes) {
    install_handler();
    if (!p) return hipErrorInvalidValue;
    if (bytes == 0) {
        *p = nullptr;
        return hipSuccess;
    }
    const size_t page = 4096;
    const size_t b16 = (bytes + 15) & ~(size_t)15;
    const size_t data = (b16 + page - 1) & ~(page - 1);
    void *base = mmap(nullptr, data + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return hipErrorOutOfMemory;
    mprotect((char *)base + data, page, PROT_NONE);
    char *ptr = (char *)base + (data - b16);
    memset(base, 0xFF, data);                       // "device" memory is never zero by accident
    std::lock_guard<std::mutex> g(g_alloc_mu);
    g_allocs[ptr] = Alloc{base, data + page, bytes};
    *p = ptr;
    return hipSuccess;
}

hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> g(g_alloc_mu);
    auto it = g_allocs.find(p);
    if (it == g_allocs.end()) {
        fprintf(stderr, "hipemu: hipFree of an unknown pointer %p\n", p);
        abort();
    }
    munmap(it->second.base, it->second.total);
    g_allocs.erase(it);
    return hipSuccess;
}

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) {
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, n, k); }
hipError_t hipMemset(void *dst, int v, size_t n) {
    if (n) memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { return hipMemset(dst, v, n); }
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t) {
    for (size_t i = 0; i < count; ++i) ((int *)dst)[i] = v;
    return hipSuccess;
}
hipError_t hipMemset2DAsync(void *dst, size_t pitch, int v, size_t width, size_t height, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memset((char *)dst + r * pitch, v, width);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    *free_b = (size_t)8 << 30;
    *total_b = (size_t)16 << 30;
    return hipSuccess;
}

// streams and events: everything runs synchronously in the emulator
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = (hipStream_t)malloc(8);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    free(s);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    *e = (hipEvent_t)malloc(8);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    free(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
