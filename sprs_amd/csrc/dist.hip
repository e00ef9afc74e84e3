// Row-sharded SpMV across the GPUs of one node, inside the library: one process per GPU, RCCL over xGMI.
//
// The reference has no distributed code; the shard is its own slice_outer (sprs/src/sparse/slicing.rs:65-89) with the
// indptr rebased (to_proper, indptr.rs:206-214): rank g owns the contiguous row block [row_starts[g], row_starts[g+1]),
// keeps a full replica of x and computes its block of y with the single-GPU kernels; ONE exchange step follows, an
// all-gather-v of y.  xGMI is a full point-to-point mesh (7 links per GPU), so the exchange is a DIRECT one — every
// rank sends its block to every peer at once, ncclGroupStart / ncclSend + ncclRecv / ncclGroupEnd — not a ring that
// would be bound by one link.  The local block is cut into sub-blocks (cost-balanced, each with its own plan): the
// sends of a finished sub-block go out on a second stream while the next sub-block is still being multiplied.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, a process that
// already carries a copy (PyTorch does) shares it, and single-GPU users never load it.
#include "common.hpp"

#include <dlfcn.h>

#include <cstring>
#include <vector>

namespace sprs_hip {

namespace {

// the few RCCL entry points used (rccl.h); ncclUniqueId is 128 opaque bytes, ncclFloat64 = 8
struct NcclId {
    char internal[128];
};
typedef void *ncclComm_t;
typedef int ncclResult_t;
constexpr int NCCL_FLOAT64 = 8;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(NcclId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, NcclId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int32_t rccl(Rccl **out) {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.lib) {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "multi-GPU: RCCL is not available (%s)", dlerror());
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
        r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
        r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv)
            SPRS_FAIL(SPRS_HIP_INVALID_ARG, "multi-GPU: librccl lacks an expected entry point");
        r.lib = h;
    }
    *out = &r;
    return SPRS_HIP_OK;
}

#define SPRS_TRY_NCCL(R, expr)                                                                                      \
    do {                                                                                                            \
        const ncclResult_t r__ = (expr);                                                                            \
        if (r__ != 0) SPRS_FAIL(SPRS_HIP_HIP_ERROR, "RCCL error %d (%s) in %s", r__,                                \
                                (R)->GetErrorString ? (R)->GetErrorString(r__) : "?", #expr);                       \
    } while (0)

// Inside ncclGroupStart / ncclGroupEnd a failing call must not return at once: the group would stay open and every later
// RCCL call of the process would be queued into it.  The first failure is kept, the group is closed, then it is reported.
struct GroupStatus {
    ncclResult_t first = 0;
    const char *what = nullptr;
    void note(ncclResult_t r, const char *expr) {
        if (r != 0 && first == 0) {
            first = r;
            what = expr;
        }
    }
};
#define SPRS_GROUP_CALL(G, expr) (G).note((expr), #expr)

int32_t close_group(Rccl *R, GroupStatus &g) {
    const ncclResult_t e = R->GroupEnd();
    g.note(e, "ncclGroupEnd");
    if (g.first != 0)
        SPRS_FAIL(SPRS_HIP_HIP_ERROR, "RCCL error %d (%s) in %s", g.first, R->GetErrorString ? R->GetErrorString(g.first) : "?", g.what);
    return SPRS_HIP_OK;
}

}  // namespace

}  // namespace sprs_hip

struct sprs_hip_dist {
    int32_t world = 1, rank = 0;
    uint64_t rows = 0, cols = 0;
    std::vector<uint64_t> row_starts;                  // world + 1
    std::vector<sprs_hip_csmat *> sub;                 // sub-blocks of the local row block (owned)
    std::vector<uint64_t> sub_starts;                  // global first row of each sub-block, + the block's end
    std::vector<std::vector<uint64_t>> peer_starts;    // the same table of every rank (exchanged at creation)
    void *comm = nullptr;
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> done;                      // sub-block s multiplied (recorded on the caller's stream)
    hipEvent_t gathered = nullptr;                     // exchange complete (recorded on comm_stream)
    // ---- the second exchange route: peer stores over xGMI (SURVEY 8e "report both") --------------------------------------
    int32_t route = 0;                                 // 0: grouped ncclSend / ncclRecv; 1: stores into the peers' receive windows
    void *window = nullptr;                            // this rank's receive window: [flags: one epoch word per peer | y copy 0 | y copy 1]
    std::vector<void *> peer_window;                   // the peers' windows, opened from their IPC handles (own entry: `window`)
    unsigned long long epoch = 0;                      // collective multiplies so far on the peer route
    int *peer_status = nullptr;                        // host-mapped: set by the wait kernel when a peer does not arrive in time
};

namespace sprs_hip {

uint64_t dist_rows(const sprs_hip_dist *d) { return d->rows; }
uint64_t dist_cols(const sprs_hip_dist *d) { return d->cols; }

// what the communicator itself says its size is (ncclCommCount): a world of one never made one
int32_t dist_comm_count(const sprs_hip_dist *d, int32_t *ranks) {
    if (!d->comm) {
        *ranks = d->world;
        return SPRS_HIP_OK;
    }
    Rccl *R = nullptr;
    SPRS_TRY(rccl(&R));
    if (!R->CommCount) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "multi-GPU: librccl lacks ncclCommCount");
    int n = 0;
    SPRS_TRY_NCCL(R, R->CommCount((ncclComm_t)d->comm, &n));
    *ranks = n;
    return SPRS_HIP_OK;
}

int32_t dist_unique_id(void *id128) {
    Rccl *R = nullptr;
    SPRS_TRY(rccl(&R));
    SPRS_TRY_NCCL(R, R->GetUniqueId((NcclId *)id128));
    return SPRS_HIP_OK;
}

void dist_free(sprs_hip_dist *d) {
    if (!d) return;
    (void)hipDeviceSynchronize();                      // nothing of this handle's exchanges may still be running
    for (size_t p = 0; p < d->peer_window.size(); ++p)
#ifndef SPRS_HIP_EMU
        if (d->peer_window[p] && (int32_t)p != d->rank) (void)hipIpcCloseMemHandle(d->peer_window[p]);
    if (d->window) (void)hipFree(d->window);
    if (d->peer_status) (void)hipHostFree(d->peer_status);
#else
        (void)p;
#endif
    for (auto *m : d->sub) sprs_hip_csmat_free(m);
    for (auto e : d->done)
        if (e) (void)hipEventDestroy(e);
    if (d->gathered) (void)hipEventDestroy(d->gathered);
    if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
    if (d->comm) {
        Rccl *R = nullptr;
        if (rccl(&R) == SPRS_HIP_OK) (void)R->CommDestroy(d->comm);
    }
    delete d;
}

// local_block: the rows [row_starts[rank], row_starts[rank + 1]) of the matrix as a CSR handle with ALL the columns and a
// zero-based indptr (what slice_outer + to_proper give).  nsub sub-blocks (>= 1) pipeline multiply and exchange.
int32_t dist_create(sprs_hip_dist **out, const void *unique_id128, int32_t world, int32_t rank, uint64_t rows, uint64_t cols,
                    const uint64_t *row_starts, const sprs_hip_csmat *local_block, int32_t nsub) {
    if (world < 1 || rank < 0 || rank >= world) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "bad world size / rank");
    if (row_starts[0] != 0 || row_starts[world] != rows) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "row_starts must run from 0 to rows");
    for (int32_t g = 0; g < world; ++g)
        if (row_starts[g] > row_starts[g + 1]) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "row_starts must be non-decreasing");
    if (local_block->storage != SPRS_HIP_CSR) SPRS_FAIL(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    if (local_block->cols != cols || local_block->rows != row_starts[rank + 1] - row_starts[rank])
        SPRS_FAIL(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    auto *d = new sprs_hip_dist();
    struct Guard {
        sprs_hip_dist *p;
        ~Guard() { dist_free(p); }
    } guard{d};
    d->world = world;
    d->rank = rank;
    d->rows = rows;
    d->cols = cols;
    d->row_starts.assign(row_starts, row_starts + world + 1);
    // ---- sub-blocks of equal cost (nnz + 8 per row, as the rank split): cut points from the block's indptr -------------
    const uint64_t lrows = local_block->rows, r0 = row_starts[rank];
    if (nsub < 1) nsub = 1;
    const int32_t nsub_req = nsub;                                 // the same on every rank: the exchange runs in nsub_req groups
    if ((uint64_t)nsub > lrows) nsub = lrows ? (int32_t)lrows : 1; // (this rank may have fewer rows than that)
    std::vector<uint64_t> cuts{0};
    if (nsub > 1) {
        std::vector<uint64_t> ip(lrows + 1);
        if (local_block->iptr_bytes == 8) {
            SPRS_TRY_HIP(hipMemcpy(ip.data(), local_block->indptr, (lrows + 1) * 8, hipMemcpyDeviceToHost));
        } else {
            std::vector<uint32_t> ip32(lrows + 1);
            SPRS_TRY_HIP(hipMemcpy(ip32.data(), local_block->indptr, (lrows + 1) * 4, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i <= lrows; ++i) ip[i] = ip32[i];
        }
        const double total = (double)ip[lrows] + 8.0 * (double)lrows;
        for (int32_t s = 1; s < nsub; ++s) {
            const double target = total * s / nsub;
            uint64_t lo = cuts.back(), hi = lrows;                 // first row r with cost(rows < r) >= target
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if ((double)ip[mid] + 8.0 * (double)mid < target) lo = mid + 1;
                else hi = mid;
            }
            cuts.push_back(lo);
        }
    }
    cuts.push_back(lrows);
    for (size_t s = 0; s + 1 < cuts.size(); ++s) {
        sprs_hip_csmat *m = nullptr;
        SPRS_TRY(slice_outer(local_block, cuts[s], cuts[s + 1], &m));
        d->sub.push_back(m);
        // an iterative caller by construction: the final plan is built now, so that the first collective multiply already runs
        // on it (and multiply #1 and #2 of a solver give the same bits; plan policy, sprs_hip.h)
        if (m->rows && m->nnz) SPRS_TRY(spmv_prepare(m, nullptr));
        d->sub_starts.push_back(r0 + cuts[s]);
    }
    d->sub_starts.push_back(r0 + lrows);
    // ---- communicator, second stream, events -------------------------------------------------------------------------------
    // (a world of ONE with an id given goes through RCCL as well — an empty exchange: what a 1-GPU box can exercise of this path)
    // world > 1 WITHOUT an id: no RCCL communicator is made; the handle can only exchange over the peer route
    // (sprs_hip_dist_peer_handle / _peer_connect / _set_route) — ranks sharing one device, or a host program that has no RCCL
    if (unique_id128) {
        Rccl *R = nullptr;
        SPRS_TRY(rccl(&R));
        NcclId id;
        memcpy(&id, unique_id128, sizeof id);
        SPRS_TRY_NCCL(R, R->CommInitRank(&d->comm, world, id, rank));
        SPRS_TRY_HIP(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
        d->done.assign(d->sub.size(), nullptr);
        for (auto &e : d->done) SPRS_TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&d->gathered, hipEventDisableTiming));
        // every rank must know where every peer cuts its block: one small grouped exchange of the sub_starts tables
        // (all ranks pass the same nsub; a rank with fewer rows than nsub pads its table with its block end)
        const size_t tab = (size_t)nsub_req + 1;
        std::vector<uint64_t> mine(tab, d->sub_starts.back());
        for (size_t i = 0; i < d->sub_starts.size() && i < tab; ++i) mine[i] = d->sub_starts[i];
        uint64_t *dev = nullptr;
        SPRS_TRY_HIP(hipMalloc((void **)&dev, (size_t)world * tab * 8));
        struct Free {
            void *p;
            ~Free() { (void)hipFree(p); }
        } fr{dev};
        SPRS_TRY_HIP(hipMemcpy(dev + (size_t)rank * tab, mine.data(), tab * 8, hipMemcpyHostToDevice));
        SPRS_TRY_NCCL(R, R->GroupStart());
        {
            GroupStatus g;
            for (int32_t p = 0; p < world; ++p) {
                if (p == rank) continue;
                SPRS_GROUP_CALL(g, R->Send(dev + (size_t)rank * tab, tab, NCCL_FLOAT64, p, d->comm, d->comm_stream));   // (8-byte words)
                SPRS_GROUP_CALL(g, R->Recv(dev + (size_t)p * tab, tab, NCCL_FLOAT64, p, d->comm, d->comm_stream));
            }
            SPRS_TRY(close_group(R, g));
        }
        SPRS_TRY_HIP(hipStreamSynchronize(d->comm_stream));
        std::vector<uint64_t> all((size_t)world * tab);
        SPRS_TRY_HIP(hipMemcpy(all.data(), dev, all.size() * 8, hipMemcpyDeviceToHost));
        d->peer_starts.resize(world);
        for (int32_t p = 0; p < world; ++p) {
            d->peer_starts[p].assign(all.begin() + (size_t)p * tab, all.begin() + (size_t)(p + 1) * tab);
            if (d->peer_starts[p].front() != row_starts[p] || d->peer_starts[p].back() != row_starts[p + 1])
                SPRS_FAIL(SPRS_HIP_INVALID_ARG, "rank %d cut its block differently from the row_starts given here", p);
        }
    }
    if (world > 1 && !d->comm_stream) {                // the peer route's own second stream and events
        SPRS_TRY_HIP(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
        d->done.assign(d->sub.size(), nullptr);
        for (auto &e : d->done) SPRS_TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        SPRS_TRY_HIP(hipEventCreateWithFlags(&d->gathered, hipEventDisableTiming));
    }
    guard.p = nullptr;
    *out = d;
    return SPRS_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The PEER route: hand-written stores over xGMI instead of ncclSend / ncclRecv.
//
// Every rank owns a RECEIVE WINDOW — one allocation: an epoch word per peer, then two copies of y (rows doubles each) — exported
// as a HIP IPC handle; the host program hands the 64-byte handles round (as it does the RCCL id) and every rank maps its peers'
// windows.  One multiply, epoch e:
//   * sub-block s is multiplied into the caller's y on `stream`; behind it, on the second stream, ONE kernel stores those rows
//     into copy e & 1 of EVERY peer's window (blockIdx.y = peer: all seven links carry data at once; a chain of
//     hipMemcpyPeerAsync on one stream would use them one after the other), and the multiply of sub-block s + 1 runs beside it;
//   * after the last sub-block a tiny kernel stores e into this rank's epoch word in every peer's window (system-scope
//     release, behind a system fence that closes the push kernels' stores);
//   * a one-wave kernel polls the own window's epoch words until every peer's reads e (system-scope acquire; a peer that
//     does not arrive within the timeout raises a status word instead of hanging the device), then the peers' rows are copied
//     from the window into the caller's y, and `stream` waits for that.
// Two copies, by parity of the epoch: a peer writes copy (e + 1) & 1 while this rank may still read copy e & 1; it writes copy
// e & 1 again only in epoch e + 2, which it cannot start before it has this rank's rows of epoch e + 1 — sent after this rank's
// copy-out of epoch e in stream order.  The windows are fine-grained device memory where the runtime offers it (remote stores
// into coarse-grained memory are not guaranteed to be seen by the local L2s before the next kernel boundary with a system-scope
// acquire); a world whose ranks share one device (tests) runs the same code through the same mappings.
// No RCCL call on this path.  Not measured on more than one GPU (none reachable from the build environment): correctness is
// covered by N ranks on one GPU, the route's speed is for the first multi-GPU run to report beside the RCCL route's.
// ---------------------------------------------------------------------------------------------------------------------------
#ifdef SPRS_HIP_EMU
// (the CPU kernel emulator has no IPC and no second device: the route exists on the hardware only)
int32_t dist_peer_handle(sprs_hip_dist *, void *) { SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: not available in the CPU emulator"); }
int32_t dist_peer_connect(sprs_hip_dist *, const void *, int32_t) { SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: not available in the CPU emulator"); }
int32_t dist_set_route(sprs_hip_dist *d, int32_t route) {
    if (route != 0) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: not available in the CPU emulator");
    d->route = 0;
    return SPRS_HIP_OK;
}
int32_t dist_route(const sprs_hip_dist *d, int32_t *route) {
    *route = d->route;
    return SPRS_HIP_OK;
}
static int32_t dist_spmv_peer(sprs_hip_dist *, const double *, double *, hipStream_t) { SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: not available in the CPU emulator"); }
#else
namespace {

constexpr int PEER_MAX = 16;
constexpr uint64_t PEER_FLAG_BYTES = 4096;             // the epoch words' share of a window (one per peer, padded)

struct PeerPtrs {
    double *dst[PEER_MAX];                             // copy e & 1 of every OTHER rank's window
    unsigned long long *flag[PEER_MAX];                // this rank's epoch word in every other rank's window
    int32_t n;
};

__global__ __launch_bounds__(256) void peer_push_kernel(const double *__restrict__ src, PeerPtrs pp, uint64_t off, uint64_t n) {
    double *dst = pp.dst[blockIdx.y] + off;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    __threadfence_system();                            // the stores have left for the peer before the kernel counts as done
}

__global__ void peer_flag_kernel(PeerPtrs pp, unsigned long long epoch) {
    __threadfence_system();
    if ((int)threadIdx.x < pp.n) __hip_atomic_store(pp.flag[threadIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void peer_wait_kernel(const unsigned long long *flags, int world, int me, unsigned long long epoch, long long timeout_ticks,
                                 int *status) {
    const int p = (int)threadIdx.x;
    if (p < world && p != me) {
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(32);
            if ((long long)wall_clock64() - t0 > timeout_ticks) {
                *status = 1 + p;
                break;
            }
        }
    }
    __threadfence_system();
}

__global__ __launch_bounds__(256) void peer_copy_out_kernel(const double *__restrict__ win, double *__restrict__ y, uint64_t own0, uint64_t own1,
                                                            uint64_t rows) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride)
        if (i < own0 || i >= own1) y[i] = win[i];
}

int32_t peer_alloc_window(sprs_hip_dist *d) {
    if (d->window) return SPRS_HIP_OK;
    if (d->world > PEER_MAX) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: at most %d ranks", PEER_MAX);
    const size_t bytes = PEER_FLAG_BYTES + 2 * (size_t)d->rows * 8 + 256;
#ifndef SPRS_HIP_EMU
    if (hipExtMallocWithFlags(&d->window, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        d->window = nullptr;
    }
#endif
    if (!d->window) SPRS_TRY_HIP(hipMalloc(&d->window, bytes));
    SPRS_TRY_HIP(hipMemset(d->window, 0, bytes));
    SPRS_TRY_HIP(hipDeviceSynchronize());
    SPRS_TRY_HIP(hipHostMalloc((void **)&d->peer_status, sizeof(int), hipHostMallocMapped));
    *d->peer_status = 0;
    return SPRS_HIP_OK;
}

}  // namespace

int32_t dist_peer_handle(sprs_hip_dist *d, void *handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands 64-byte handles round");
    SPRS_TRY(peer_alloc_window(d));
    hipIpcMemHandle_t h;
    SPRS_TRY_HIP(hipIpcGetMemHandle(&h, d->window));
    memcpy(handle64, &h, 64);
    return SPRS_HIP_OK;
}

int32_t dist_peer_connect(sprs_hip_dist *d, const void *handles, int32_t world) {
    if (world != d->world) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: %d handles for a world of %d", world, d->world);
    SPRS_TRY(peer_alloc_window(d));
    if (!d->peer_window.empty()) return SPRS_HIP_OK;   // connected already
    d->peer_window.assign(world, nullptr);
    d->peer_window[d->rank] = d->window;
    for (int32_t p = 0; p < world; ++p) {
        if (p == d->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)p * 64, 64);
        SPRS_TRY_HIP(hipIpcOpenMemHandle(&d->peer_window[p], h, hipIpcMemLazyEnablePeerAccess));
    }
    return SPRS_HIP_OK;
}

int32_t dist_set_route(sprs_hip_dist *d, int32_t route) {
    if (route != 0 && route != 1) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "route must be 0 (RCCL) or 1 (peer stores)");
    if (route == 0 && d->world > 1 && !d->comm) SPRS_FAIL(SPRS_HIP_INVALID_ARG, "this handle was made without an RCCL id: only the peer route is available");
    if (route == 1 && d->world > 1 && d->peer_window.empty())
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "peer route: exchange the window handles first (sprs_hip_dist_peer_handle on every rank, sprs_hip_dist_peer_connect with all of them)");
    d->route = route;
    return SPRS_HIP_OK;
}

int32_t dist_route(const sprs_hip_dist *d, int32_t *route) {
    *route = d->route;
    return SPRS_HIP_OK;
}

static int32_t dist_spmv_peer(sprs_hip_dist *d, const double *x, double *y, hipStream_t stream) {
    if (d->peer_status && *d->peer_status) {
        const int who = *d->peer_status - 1;
        *d->peer_status = 0;
        SPRS_FAIL(SPRS_HIP_HIP_ERROR, "peer route: rank %d did not deliver its rows of the previous multiply in time", who);
    }
    const unsigned long long e = ++d->epoch;
    const uint64_t copy_off = PEER_FLAG_BYTES / 8 + (e & 1ull) * d->rows;           // in doubles
    PeerPtrs pp;
    pp.n = 0;
    for (int32_t p = 0; p < d->world; ++p) {
        if (p == d->rank) continue;
        pp.dst[pp.n] = (double *)d->peer_window[p] + copy_off;
        pp.flag[pp.n] = (unsigned long long *)d->peer_window[p] + d->rank;
        ++pp.n;
    }
    for (size_t s = 0; s < d->sub.size(); ++s) {
        sprs_hip_csmat *m = d->sub[s];
        const uint64_t r0 = d->sub_starts[s], n = d->sub_starts[s + 1] - r0;
        if (m->rows) SPRS_TRY(spmv_f64(m, x, y + r0, false, stream));
        SPRS_TRY_HIP(hipEventRecord(d->done[s], stream));
        SPRS_TRY_HIP(hipStreamWaitEvent(d->comm_stream, d->done[s], 0));
        if (n && pp.n) {
            uint64_t blocks = (n + 255) / 256;
            if (blocks > 512) blocks = 512;                                        // per peer: 7 x 512 workgroups keep every link busy
            hipLaunchKernelGGL(peer_push_kernel, dim3((unsigned)blocks, (unsigned)pp.n), dim3(256), 0, d->comm_stream, (const double *)(y + r0), pp, r0, n);
            SPRS_TRY_HIP(hipGetLastError());
        }
    }
    if (pp.n) {
        hipLaunchKernelGGL(peer_flag_kernel, dim3(1), dim3(64), 0, d->comm_stream, pp, e);
        SPRS_TRY_HIP(hipGetLastError());
        // 2 s at the 100 MHz wall clock: a peer that died must not hang the device (and the box with it)
        hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, d->comm_stream, (const unsigned long long *)d->window, d->world, d->rank, e,
                           200000000ll, d->peer_status);
        SPRS_TRY_HIP(hipGetLastError());
        uint64_t blocks = (d->rows + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(peer_copy_out_kernel, dim3((unsigned)blocks), dim3(256), 0, d->comm_stream, (const double *)d->window + copy_off, y,
                           d->row_starts[d->rank], d->row_starts[d->rank + 1], d->rows);
        SPRS_TRY_HIP(hipGetLastError());
    }
    SPRS_TRY_HIP(hipEventRecord(d->gathered, d->comm_stream));
    SPRS_TRY_HIP(hipStreamWaitEvent(stream, d->gathered, 0));
    return SPRS_HIP_OK;
}

#endif

// y (full length, on this rank's device) = A * x: the own block is multiplied sub-block by sub-block on `stream`; as soon
// as one is done its rows go to every peer on the second stream, and the peers' blocks arrive into y; `stream` then waits
// for the exchange.  Every rank must call this with the same sequence of calls (collective).
int32_t dist_spmv(sprs_hip_dist *d, const double *x, double *y, hipStream_t stream) {
    if (d->world > 1 && d->route == 1) return dist_spmv_peer(d, x, y, stream);
    if (d->world > 1 && !d->comm)
        SPRS_FAIL(SPRS_HIP_INVALID_ARG, "this handle was made without an RCCL id: connect the peer route first (sprs_hip_dist_peer_connect, sprs_hip_dist_set_route)");
    Rccl *R = nullptr;
    if (d->comm) SPRS_TRY(rccl(&R));
    const size_t groups = d->comm ? d->peer_starts[0].size() - 1 : d->sub.size();
    for (size_t s = 0; s < groups; ++s) {
        if (s < d->sub.size()) {
            sprs_hip_csmat *m = d->sub[s];
            if (m->rows) SPRS_TRY(spmv_f64(m, x, y + d->sub_starts[s], false, stream));
        }
        if (!d->comm) continue;
        // sub-block s is done: its rows go to every peer, the peers' sub-block s arrives — on the second stream, while the
        // caller's stream goes on with sub-block s + 1
        const size_t ev = s < d->done.size() ? s : d->done.size() - 1;
        SPRS_TRY_HIP(hipEventRecord(d->done[ev], stream));
        SPRS_TRY_HIP(hipStreamWaitEvent(d->comm_stream, d->done[ev], 0));
        const std::vector<uint64_t> &me = d->peer_starts[d->rank];
        SPRS_TRY_NCCL(R, R->GroupStart());
        GroupStatus g;
        for (int32_t p = 0; p < d->world; ++p) {
            if (p == d->rank) continue;
            const std::vector<uint64_t> &pe = d->peer_starts[p];
            if (me[s + 1] > me[s]) SPRS_GROUP_CALL(g, R->Send(y + me[s], me[s + 1] - me[s], NCCL_FLOAT64, p, d->comm, d->comm_stream));
            if (pe[s + 1] > pe[s]) SPRS_GROUP_CALL(g, R->Recv(y + pe[s], pe[s + 1] - pe[s], NCCL_FLOAT64, p, d->comm, d->comm_stream));
        }
        SPRS_TRY(close_group(R, g));
    }
    if (d->comm) {
        SPRS_TRY_HIP(hipEventRecord(d->gathered, d->comm_stream));
        SPRS_TRY_HIP(hipStreamWaitEvent(stream, d->gathered, 0));      // y is complete for whatever the caller queues next
    }
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
