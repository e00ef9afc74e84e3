#!/usr/bin/env python3
"""Why does the SpMM object of bench.py time the k = 16 call at 7.3 ms when scripts/spmm_bench.py times it at 5.8 ms in the same session?
The same call under different circumstances: fresh buffers / buffers carved from torch's cache / after an idle pause / more warm-up calls."""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402


def main():
    n, k = 10_000_000, 16
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)

    def measure(tag, warm=2, reps=10, k=16):
        rhs = gen.dense_vector(n * k, seed=5, device=dev)
        out = torch.empty(n * k, dtype=torch.float64, device=dev)
        call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(a._h, C.c_void_p(rhs.data_ptr()), n, k, k, C.c_void_p(out.data_ptr()), n, k, 0, None))
        for _ in range(warm):
            call()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        t0 = time.perf_counter()
        for p, q in evs:
            p.record()
            call()
            q.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        per = [round(p.elapsed_time(q), 3) for p, q in evs]
        print(json.dumps({"case": tag, "warmup": warm, "wall_ms": round(wall, 3), "event_ms": per, "k": k, "rhs_ptr": hex(rhs.data_ptr()), "out_ptr": hex(out.data_ptr())}), flush=True)

    def ptrs():
        return {"indices": hex(indices.data_ptr()), "data": hex(data.data_ptr())}

    print(json.dumps(ptrs()))
    mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
    if mode == "k8first":                                   # what scripts/spmm_bench.py does before its k = 16 line
        measure("k = 8 first", k=8)
        measure("then k = 16")
    elif mode == "like_bench":                              # spmm_bench.py's sequence: k = 8, an SpMV on the handle, then k = 16
        from sprs_amd import prod
        from sprs_amd.device import DeviceVec
        measure("k = 8 first", k=8)
        x0 = gen.dense_vector(n, seed=3, device=dev)
        y0 = torch.empty(n, dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(x0), out=DeviceVec.borrow(y0))
        torch.cuda.synchronize()
        measure("k = 16 after k = 8 and an SpMV")
        measure("k = 16 again")
        measure("k = 8 again", k=8)
        measure("k = 16 a third time")
    elif mode == "spmvplan":                                # the banded SpMV plan (3.5 GB of library allocations) built first
        from sprs_amd import prod
        from sprs_amd.device import DeviceVec
        x0 = gen.dense_vector(n, seed=3, device=dev)
        y0 = torch.empty(n, dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(x0), out=DeviceVec.borrow(y0))
        torch.cuda.synchronize()
        measure("after the SpMV plan was built")
    elif mode == "aligned":                                 # rhs and out from ONE fresh allocation each, 2 MiB aligned by construction
        torch.cuda.empty_cache()
        measure("fresh, cache emptied first")
    else:
        measure("fresh buffers")


if __name__ == "__main__":
    main()
