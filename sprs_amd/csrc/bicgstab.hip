// Device-resident BiCGSTAB: twin of sprs::linalg::bicgstab::BiCGSTAB::solve
// (sprs/src/sparse/linalg/bicgstab.rs:148-171; new() :117-143, step() :194-229, soft/hard
// restart :175-192) — SURVEY §8 (f3), "callers that loop on SpMV".
//
// The reference keeps x, r, rhat, p as CsVec and forms every product with `&CsMat * &CsVec`; for
// vectors without structural zeros that is the dense arithmetic done here (csr_mul_csvec,
// prod.rs:162-184, is one ascending sparse dot per row; a CSC operand takes the SpGEMM route, which
// also adds ascending k).  Everything stays in HBM: per iteration two SpMVs (the
// hot path of this library), four fused element-wise kernels and three dot launches; the host sees
// five scalars per iteration (alpha, omega, rho need them anyway: the restart decisions are
// data dependent, bicgstab.rs:219-226).
//
// Arithmetic: unfused (-ffp-contract=off), same operand order as the reference's expressions, so
// the element-wise part is bit-identical.  (The SpMV is within its own 1e-10 bar: a row that straddles
// two nnz tiles is summed as tail + head.)  Dot products / norms: the reference sums serially from
// 0 (vec.rs:846-880, 907-918).  For n <= BICG_SERIAL_N one thread does exactly that — the
// reference's own test system (4 x 4, tol 1e-60, bicgstab.rs:336-369) then converges to a residual of
// exactly zero in 45 iterations, as a serial CPU restatement of the reference does; above it a FIXED
// two-level tree (8192-element chunks, 256 threads each, partials summed by one workgroup) —
// deterministic, independent of the
// launch, but rounded differently from a serial sum: iterates agree with the serial ones to
// rounding and the restart counts may differ.
#include "common.hpp"

#include <cmath>

namespace sprs_hip {

namespace {

constexpr uint64_t BICG_SERIAL_N = 2048;
constexpr int DOT_BLOCK = 256;
constexpr uint64_t DOT_CHUNK = 8192;

// fixed-shape block reduction: lane tree inside the wave, then the waves in order
__device__ __forceinline__ double block_sum_fixed(double v, double *lds /* >= 4 */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        s = lds[0];
        for (uint32_t w = 1; w < (uint32_t)(DOT_BLOCK / 64); ++w) s += lds[w];
    }
    __syncthreads();
    return s;   // valid in thread 0
}

// out[0] = sum a_i b_i, out[1] = sum c_i d_i (second pair optional), serial order: n <= BICG_SERIAL_N
__global__ void dot_serial_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                  const double *__restrict__ c, const double *__restrict__ d, uint64_t n,
                                  double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s0 = 0.0, s1 = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        const double p0 = a[i] * b[i];
        s0 = s0 + p0;
        if (c) {
            const double p1 = c[i] * d[i];
            s1 = s1 + p1;
        }
    }
    out[0] = s0;
    out[1] = s1;
}

// stage 1: chunk c = [c * DOT_CHUNK, ...): thread t sums elements t, t + 256, ... serially, then the block tree
__global__ __launch_bounds__(DOT_BLOCK) void dot_partial_kernel(const double *__restrict__ a,
                                                                const double *__restrict__ b,
                                                                const double *__restrict__ c,
                                                                const double *__restrict__ d, uint64_t n,
                                                                double *__restrict__ partial /* 2 x nchunks */) {
    __shared__ double lds[8];
    const uint64_t lo = (uint64_t)blockIdx.x * DOT_CHUNK;
    const uint64_t hi = lo + DOT_CHUNK < n ? lo + DOT_CHUNK : n;
    double s0 = 0.0, s1 = 0.0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += DOT_BLOCK) {
        const double p0 = a[i] * b[i];
        s0 = s0 + p0;
        if (c) {
            const double p1 = c[i] * d[i];
            s1 = s1 + p1;
        }
    }
    const double t0 = block_sum_fixed(s0, lds);
    const double t1 = block_sum_fixed(s1, lds + 4);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = t0;
        partial[gridDim.x + blockIdx.x] = t1;
    }
}

// stage 2: one workgroup, thread t sums partials t, t + 256, ... serially, then the block tree
__global__ __launch_bounds__(DOT_BLOCK) void dot_final_kernel(const double *__restrict__ partial, uint64_t nchunks,
                                                              double *__restrict__ out) {
    __shared__ double lds[8];
    double s0 = 0.0, s1 = 0.0;
    for (uint64_t i = threadIdx.x; i < nchunks; i += DOT_BLOCK) {
        s0 = s0 + partial[i];
        s1 = s1 + partial[nchunks + i];
    }
    const double t0 = block_sum_fixed(s0, lds);
    const double t1 = block_sum_fixed(s1, lds + 4);
    if (threadIdx.x == 0) {
        out[0] = t0;
        out[1] = t1;
    }
}

// ---- element-wise steps, written as the reference writes them --------------------------------
// new(): r = b - A x0 (v holds A x0); rhat = r; p = r; x = x0      (bicgstab.rs:123-128)
__global__ void init_kernel(const double *__restrict__ b, const double *__restrict__ v, const double *__restrict__ x0,
                            uint64_t n, double *__restrict__ r, double *__restrict__ rhat, double *__restrict__ p,
                            double *__restrict__ x) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ri = b[i] - v[i];
    r[i] = ri;
    rhat[i] = ri;
    p[i] = ri;
    x[i] = x0[i];
}

// h = x + p * alpha;  s = r - v * alpha                            (bicgstab.rs:200, 203)
__global__ void hs_kernel(const double *__restrict__ x, const double *__restrict__ p, const double *__restrict__ r,
                          const double *__restrict__ v, double alpha, uint64_t n, double *__restrict__ h,
                          double *__restrict__ s) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double pa = p[i] * alpha;
    h[i] = x[i] + pa;
    const double va = v[i] * alpha;
    s[i] = r[i] - va;
}

// x = h + omega * s;  r = s - t * omega                            (bicgstab.rs:206, 209)
__global__ void xr_kernel(const double *__restrict__ h, const double *__restrict__ s, const double *__restrict__ t,
                          double omega, uint64_t n, double *__restrict__ x, double *__restrict__ r) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double os = omega * s[i];
    x[i] = h[i] + os;
    const double to = t[i] * omega;
    r[i] = s[i] - to;
}

// p = r + (p - v * omega) * beta                                   (bicgstab.rs:224-226)
__global__ void p_kernel(const double *__restrict__ r, const double *__restrict__ v, double omega, double beta,
                         uint64_t n, double *__restrict__ p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double vo = v[i] * omega;
    const double d = p[i] - vo;
    const double e = d * beta;
    p[i] = r[i] + e;
}

// soft_restart(): rhat = r; p = r                                  (bicgstab.rs:175-180)
__global__ void restart_kernel(const double *__restrict__ r, uint64_t n, double *__restrict__ rhat,
                               double *__restrict__ p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rhat[i] = r[i];
    p[i] = r[i];
}

// hard_restart(): r = b - A x (v holds A x)                        (bicgstab.rs:186)
__global__ void resid_kernel(const double *__restrict__ b, const double *__restrict__ v, uint64_t n,
                             double *__restrict__ r) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    r[i] = b[i] - v[i];
}

struct Work {
    double *buf = nullptr;
    ~Work() {
        if (buf) (void)hipFree(buf);
    }
};

}  // namespace

int32_t bicgstab_f64(sprs_hip_csmat *a_in, const double *x0, const double *b, uint64_t n, double tol, uint64_t max_iter,
                     double soft_restart_threshold, double *x, sprs_hip_bicgstab_info *info, hipStream_t stream) {
    // `&a * &x` of a CSC matrix (csmat.rs:1866-1949 route) == CSR SpMV of its CSR form: convert once
    sprs_hip_csmat *a = a_in, *converted = nullptr;
    if (a_in->storage != SPRS_HIP_CSR) {
        SPRS_TRY(to_other_storage(a_in, &converted));
        a = converted;
    }
    struct Guard {
        sprs_hip_csmat *m;
        ~Guard() {
            if (m) sprs_hip_csmat_free(m);
        }
    } guard{converted};

    const uint64_t nchunks = (n + DOT_CHUNK - 1) / DOT_CHUNK;
    Work w;
    const uint64_t doubles = 7 * n + 2 * nchunks + 8;
    SPRS_TRY_HIP(hipMalloc((void **)&w.buf, doubles * sizeof(double)));
    double *r = w.buf, *rhat = r + n, *p = rhat + n, *v = p + n, *s = v + n, *t = s + n, *h = t + n;
    double *partial = h + n, *scal = partial + 2 * nchunks;
    const dim3 eg((unsigned)((n + 255) / 256)), eb(256);

    // two dots in one go; returns them on the host (the stream is drained: the next scalar depends on it)
    auto dots = [&](const double *a0, const double *b0, const double *c0, const double *d0, double &o0,
                    double &o1) -> int32_t {
        if (n <= BICG_SERIAL_N) {
            hipLaunchKernelGGL(dot_serial_kernel, dim3(1), dim3(64), 0, stream, a0, b0, c0, d0, n, scal);
        } else {
            hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)nchunks), dim3(DOT_BLOCK), 0, stream, a0, b0, c0, d0, n,
                               partial);
            hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(DOT_BLOCK), 0, stream, partial, nchunks, scal);
        }
        SPRS_TRY_HIP(hipGetLastError());
        double hst[2];
        SPRS_TRY_HIP(hipMemcpyAsync(hst, scal, 16, hipMemcpyDeviceToHost, stream));
        SPRS_TRY_HIP(hipStreamSynchronize(stream));
        o0 = hst[0];
        o1 = hst[1];
        return SPRS_HIP_OK;
    };
    const double *none = nullptr;
    double d0 = 0.0, d1 = 0.0;

    // ---- new() ---------------------------------------------------------------------------------
    SPRS_TRY(spmv_f64(a, x0, v, false, stream));
    hipLaunchKernelGGL(init_kernel, eg, eb, 0, stream, b, v, x0, n, r, rhat, p, x);
    SPRS_TRY_HIP(hipGetLastError());
    SPRS_TRY(dots(r, r, none, none, d0, d1));
    double err = std::sqrt(d0);
    double rho = err * err;
    uint64_t it = 0, soft = 0, hard = 0;
    int32_t converged = 0;

    for (uint64_t k = 0; k < max_iter && !converged; ++k) {
        // ---- step() ----------------------------------------------------------------------------
        ++it;
        SPRS_TRY(spmv_f64(a, p, v, false, stream));
        SPRS_TRY(dots(rhat, v, none, none, d0, d1));
        const double alpha = rho / d0;
        hipLaunchKernelGGL(hs_kernel, eg, eb, 0, stream, x, p, r, v, alpha, n, h, s);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY(spmv_f64(a, s, t, false, stream));
        SPRS_TRY(dots(t, s, t, t, d0, d1));
        const double omega = d0 / d1;
        hipLaunchKernelGGL(xr_kernel, eg, eb, 0, stream, h, s, t, omega, n, x, r);
        SPRS_TRY_HIP(hipGetLastError());
        SPRS_TRY(dots(r, r, rhat, r, d0, d1));
        err = std::sqrt(d0);
        const double rho_prev = rho;
        rho = d1;
        if (std::fabs(rho) / (err * err) < soft_restart_threshold) {
            ++soft;
            hipLaunchKernelGGL(restart_kernel, eg, eb, 0, stream, r, n, rhat, p);
            rho = err * err;
        } else {
            const double beta = (rho / rho_prev) * (alpha / omega);
            hipLaunchKernelGGL(p_kernel, eg, eb, 0, stream, r, v, omega, beta, n, p);
        }
        SPRS_TRY_HIP(hipGetLastError());
        // ---- solve(): check the TRUE error before claiming convergence ---------------------------
        if (err < tol) {
            ++hard;
            SPRS_TRY(spmv_f64(a, x, v, false, stream));
            hipLaunchKernelGGL(resid_kernel, eg, eb, 0, stream, b, v, n, r);
            SPRS_TRY_HIP(hipGetLastError());
            SPRS_TRY(dots(r, r, none, none, d0, d1));
            err = std::sqrt(d0);
            hipLaunchKernelGGL(restart_kernel, eg, eb, 0, stream, r, n, rhat, p);
            SPRS_TRY_HIP(hipGetLastError());
            rho = err * err;
            if (err < tol) converged = 1;
        }
    }
    SPRS_TRY_HIP(hipStreamSynchronize(stream));
    if (info) {
        info->iteration_count = it;
        info->soft_restart_count = soft;
        info->hard_restart_count = hard;
        info->err = err;
        info->rho = rho;
        info->converged = converged;
    }
    return SPRS_HIP_OK;
}

}  // namespace sprs_hip
