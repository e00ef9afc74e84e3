#!/usr/bin/env python3
"""Time per BiCGSTAB iteration on the device (sprs_hip_bicgstab_f64) against its parts: two SpMVs of
the same handle plus the vector traffic of one iteration (23 n-vectors of 8 bytes read or written).
usage: bicgstab_bench.py [grid]   (5-point Laplacian of heat.rs on a grid x grid mesh, default 4096)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import gen, prod                       # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec   # noqa: E402
from sprs_amd.linalg import BiCGSTAB                 # noqa: E402


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda", 0)
    n = g * g
    indptr, indices, data = gen.grid_laplacian(g, g, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    b = DeviceVec.borrow(gen.dense_vector(n, seed=3, device=dev))
    x0 = DeviceVec.borrow(torch.zeros(n, dtype=torch.float64, device=dev))
    BiCGSTAB.solve(a, x0, b, 1e-300, 2)                       # warm-up: plan, allocations
    torch.cuda.synchronize()
    iters = 40
    t0 = time.perf_counter()
    res = BiCGSTAB.solve(a, x0, b, 1e-300, iters)             # tolerance out of reach: exactly `iters` steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert res.iteration_count() == iters
    y = DeviceVec.borrow(torch.zeros(n, dtype=torch.float64, device=dev))
    for _ in range(3):
        prod.csmat_mul_vec(a, b, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        prod.csmat_mul_vec(a, b, out=y)
    torch.cuda.synchronize()
    spmv = (time.perf_counter() - t0) / 20
    per_it = dt / iters
    vec_bytes = 23 * n * 8
    out = {"matrix": "5-pt Laplacian %d^2" % g, "n": n, "nnz": int(indices.numel()), "iterations": iters,
           "ms_per_iteration": round(per_it * 1e3, 4), "spmv_ms": round(spmv * 1e3, 4),
           "two_spmv_share": round(2 * spmv / per_it, 3),
           "vector_bytes_per_iteration": vec_bytes,
           "vector_part_GBps": round(vec_bytes / max(per_it - 2 * spmv, 1e-9) / 1e9, 1),
           "err_after": res.err(), "soft_restarts": res.soft_restart_count()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
