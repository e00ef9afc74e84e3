#!/usr/bin/env python3
"""What bounds the SpMM kernel: the random fills of rhs rows, or the dependent loads of its row walk?  The same kernel on three
matrices of 10M rows x ~32 entries: (a) columns drawn from 32768 (the rhs fits every L2: no fills), (b) columns uniform over 10M
(every rhs row a fill from HBM, no reuse), (c) the bench's R-MAT.  (a) ~ (b): the walk is the bound; (a) << (b): the fills are.
usage: spmm_bound_probe.py [k ...]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402


def uniform_rows(n, cols, per_row, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    c = torch.randint(0, cols - per_row, (n, per_row), device=dev, generator=g, dtype=torch.int64)
    c = torch.sort(c, dim=1).values + torch.arange(per_row, device=dev)        # strictly increasing in every row
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * per_row
    data = gen.dense_vector(n * per_row, seed=seed + 1, device=dev)
    return indptr, c.reshape(-1).contiguous(), data


def run(name, shape, indptr, indices, data, ks, dev):
    rows, cols = shape
    a = DeviceCsMat.wrap_torch(shape, indptr, indices, data)
    for k in ks:
        rhs = gen.dense_vector(cols * k, seed=5, device=dev)
        out = torch.empty(rows * k, dtype=torch.float64, device=dev)
        call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(
            a._h, C.c_void_p(rhs.data_ptr()), cols, k, k, C.c_void_p(out.data_ptr()), rows, k, 0, None))
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        nnz = indices.numel()
        print(json.dumps({"matrix": name, "rows": rows, "cols": cols, "nnz": nnz, "k": k, "ms": round(dt * 1e3, 3),
                          "ps_per_entry": round(dt / nnz * 1e12, 1), "rhs_gather_TBs": round(nnz * k * 8 / dt / 1e12, 2)}), flush=True)
        del rhs, out


def main():
    ks = [int(v) for v in sys.argv[1:]] or [8, 16]
    dev = torch.device("cuda", 0)
    n = 10_000_000
    for opt in (os.environ.get("SPRS_OPTS") or "").split():
        import sprs_amd
        key, val = opt.split("=")
        sprs_amd.set_option(key, int(val))
    run("uniform32_cols32768", (n, 32768), *uniform_rows(n, 32768, 32, dev, 11), ks, dev)
    run("uniform32_cols10M", (n, n), *uniform_rows(n, n, 32, dev, 13), ks, dev)
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    run("rmat10m", (n, n), indptr, indices, data, ks, dev)


if __name__ == "__main__":
    main()
