#!/usr/bin/env python3
"""What would XCD-affine SpMM be worth?  Probe with the EXISTING kernel: keep only the entries whose rhs row hashes to one of the
8 x-line slices (x_slice of spmv_shared.hpp), multiply that eighth of the matrix — every XCD's L2 then sees the same eighth of
the rhs rows, which is the hit rate an XCD-affine kernel would have — and compare 8 x that time with the whole matrix.
usage: spmm_affinity_probe.py [k=16]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402
import ctypes as C                                   # noqa: E402


def time_spmm(a, n, k, rhs, out, reps=6):
    call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(
        a._h, C.c_void_p(rhs.data_ptr()), n, k, k, C.c_void_p(out.data_ptr()), n, k, 0, None))
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = 10_000_000
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    rhs = gen.dense_vector(n * k, seed=5, device=dev)
    out = torch.empty(n * k, dtype=torch.float64, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    whole = time_spmm(a, n, k, rhs, out)
    rows_of = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]))
    h = ((indices >> 4) * (-7046029254386353131)) >> 61 & 7          # top 3 bits of (line * 0x9E3779B97F4A7C15) mod 2^64
    res = {"k": k, "whole_ms": round(whole * 1e3, 3), "slices_ms": []}
    for s in (0, 3):
        keep = h == s
        ix, dt, ro = indices[keep].contiguous(), data[keep].contiguous(), rows_of[keep]
        ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ip[1:] = torch.cumsum(torch.bincount(ro, minlength=n), 0)
        b = DeviceCsMat.wrap_torch((n, n), ip, ix, dt)
        t = time_spmm(b, n, k, rhs, out)
        res["slices_ms"].append({"slice": s, "nnz": int(ix.numel()), "ms": round(t * 1e3, 3)})
        del b, ix, dt, ro, ip
    res["eight_slices_ms_estimate"] = round(8 * sum(x["ms"] for x in res["slices_ms"]) / len(res["slices_ms"]), 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
