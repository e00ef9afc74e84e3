#!/usr/bin/env python3
"""Triplet (COO) -> CSR assembly at scale on the device (sprs_hip_triplets_to_cs, triplet.hip): n random triplets of an
R-MAT-like shape generated on the GPU (duplicates included), timed through the C ABI with device-resident inputs, and the
selector-product route (to_other_storage + mul_csr_csr, the round-1 implementation) beside it on a smaller sample.
Algorithmic bytes of the sort route: per triplet 8 (row) + 8 (col) + 8 (value) read, 16 written (index + value) once,
plus the radix passes (32 B per triplet and pass; sort.hip).
usage: triplet_bench.py [n_triplets] [rows]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi                           # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    # skewed rows / columns (squares of uniforms), so that duplicates and long rows exist
    r = (torch.rand(n, device=dev, generator=g) ** 2 * rows).long().clamp_(max=rows - 1)
    c = (torch.rand(n, device=dev, generator=g) ** 2 * rows).long().clamp_(max=rows - 1)
    v = torch.rand(n, device=dev, generator=g, dtype=torch.float64) - 0.5
    torch.cuda.synchronize()
    times = []
    nnz = 0
    for _ in range(3):
        h = C.c_void_p()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _ffi.check(_ffi.lib.sprs_hip_triplets_to_cs(rows, rows, n, C.c_void_p(r.data_ptr()), C.c_void_p(c.data_ptr()), 8,
                                                    C.c_void_p(v.data_ptr()), _ffi.CSR, 8, 8, C.byref(h)))
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        m = DeviceCsMat(h.value)
        nnz = m.nnz()
        del m
    out = {"triplets": n, "rows": rows, "nnz_out": int(nnz), "seconds_best": round(min(times), 4), "seconds_all": [round(t, 4) for t in times],
           "Mtriplets_per_s": round(n / min(times) / 1e6, 1)}
    # value check: total of the values is kept by the assembly (sums of duplicates), up to rounding
    h = C.c_void_p()
    _ffi.check(_ffi.lib.sprs_hip_triplets_to_cs(rows, rows, n, C.c_void_p(r.data_ptr()), C.c_void_p(c.data_ptr()), 8,
                                                C.c_void_p(v.data_ptr()), _ffi.CSR, 8, 8, C.byref(h)))
    m = DeviceCsMat(h.value)
    p_ip, p_ix, p_dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _ffi.check(_ffi.lib.sprs_hip_csmat_device_ptrs(m._h, C.byref(p_ip), C.byref(p_ix), C.byref(p_dt)))
    dt = torch.empty(nnz, dtype=torch.float64, device=dev)
    _ffi.check(_ffi.lib.sprs_hip_memcpy_d2d(C.c_void_p(dt.data_ptr()), p_dt, nnz * 8, None))
    torch.cuda.synchronize()
    out["sum_in"] = float(v.sum())
    out["sum_out"] = float(dt.sum())
    del m, dt
    # selector-product route on a sample (it goes through the SpGEMM: n columns of R, n rows of E)
    ns = min(n, 10_000_000)
    from sprs_amd.triplet import TriMat
    tm = TriMat((rows, rows), r[:ns].cpu().numpy(), c[:ns].cpu().numpy(), v[:ns].cpu().numpy())
    t0 = time.perf_counter()
    a = tm.to_csr(method="product")
    torch.cuda.synchronize()
    out["selector_product_route"] = {"triplets": ns, "seconds_incl_upload": round(time.perf_counter() - t0, 4), "nnz_out": a.nnz()}
    t0 = time.perf_counter()
    b = tm.to_csr(method="sort")
    torch.cuda.synchronize()
    out["sort_route_same_sample"] = {"triplets": ns, "seconds_incl_upload": round(time.perf_counter() - t0, 4), "nnz_out": b.nnz()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
