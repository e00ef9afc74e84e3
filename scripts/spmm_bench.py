#!/usr/bin/env python3
"""SpMM (csr_mulacc_dense_rowmaj twin) timing on the bench matrix: R-MAT n x n, rhs n x k.
Algorithmic bytes per SpMM = nnz*(8+S_I) + (rows+1)*S_P + cols*k*8 [rhs once] + rows*k*8 [out].
usage: spmm_bench.py [n] [nnz_per_row] [k ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402
import ctypes as C                                   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nzr = float(sys.argv[2]) if len(sys.argv) > 2 else 32
    ks = [int(v) for v in sys.argv[3:]] or [8, 16, 32]
    dev = torch.device("cuda", 0)
    if os.environ.get("SPMM_LONG_ROW"):
        import sprs_amd
        sprs_amd.set_option("spmm_long_row", int(os.environ["SPMM_LONG_ROW"]))
    for opt in (os.environ.get("SPRS_OPTS") or "").split():
        import sprs_amd
        key, val = opt.split("=")
        sprs_amd.set_option(key, int(val))
    indptr, indices, data = gen.rmat_csr(n, nzr, device=dev)
    if os.environ.get("PERMUTE_COLS"):
        # the same sparsity statistics with the hub columns no longer at 0, 2^k, 2^j + 2^k: separates what the ADDRESSES of the hot
        # rhs rows cost from what the power law costs (as bench.py --permute-cols does for the SpMV)
        g = torch.Generator(device=dev)
        g.manual_seed(int(os.environ["PERMUTE_COLS"]))
        perm = torch.randperm(n, device=dev, generator=g).to(indices.dtype)
        if os.environ["PERMUTE_COLS"] == "0":                    # control: the same allocations and kernels, columns left where they are
            perm = torch.arange(n, device=dev, dtype=indices.dtype)
        indices = perm[indices.long()]
        rows_of = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
        key, order = torch.sort((rows_of << 32) | indices.long())
        indices = (key & 0xFFFFFFFF).to(indices.dtype)
        data = data[order]
        del perm, rows_of, key, order
        torch.cuda.empty_cache()
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    nnz = indices.numel()
    for k in ks:
        rhs = gen.dense_vector(n * k, seed=5, device=dev)
        out = torch.empty(n * k, dtype=torch.float64, device=dev)
        call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(
            a._h, C.c_void_p(rhs.data_ptr()), n, k, k, C.c_void_p(out.data_ptr()), n, k, 0, None))
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        alg = nnz * 16 + (n + 1) * 8 + 2 * n * k * 8
        # parity of column 0 against the SpMV path (itself checked against the oracle)
        from sprs_amd import prod
        from sprs_amd.device import DeviceVec
        variant = os.environ.get("SPMM_BENCH_VARIANT", "")       # (bisecting what makes the k = 16 call fast after k = 8: DESIGN 4.3)
        err = None
        if variant == "free_first":
            x0 = rhs.view(n, k)[:, 0].contiguous()
            del rhs, out
            rhs = out = None
            y0 = torch.empty(n, dtype=torch.float64, device=dev)
            prod.csmat_mul_vec(a, DeviceVec.borrow(x0), out=DeviceVec.borrow(y0))
            torch.cuda.synchronize()
        elif variant == "no_spmv":
            x0 = rhs.view(n, k)[:, 0].contiguous()
            y0 = torch.empty(n, dtype=torch.float64, device=dev)
            got = out.view(n, k)[:, 0]
            err = float(((got - y0).abs() / y0.abs().clamp_min(1e-300)).max())
        elif variant != "no_parity":
            x0 = rhs.view(n, k)[:, 0].contiguous()
            y0 = torch.empty(n, dtype=torch.float64, device=dev)
            prod.csmat_mul_vec(a, DeviceVec.borrow(x0), out=DeviceVec.borrow(y0))
            torch.cuda.synchronize()
            got = out.view(n, k)[:, 0]
            err = float(((got - y0).abs() / y0.abs().clamp_min(1e-300)).max())
        print(json.dumps({"n": n, "nnz": nnz, "k": k, "ms": round(dt * 1e3, 4), "gflops": round(2 * nnz * k / dt / 1e9, 1),
                          "algorithmic_GBs": round(alg / dt / 1e9, 1), "frac_of_8TBs": round(alg / dt / 8e12, 4),
                          "col0_vs_spmv_max_rel": err}))
        del rhs, out


if __name__ == "__main__":
    main()
