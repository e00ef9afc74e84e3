#!/usr/bin/env python3
"""SpMM k = 16 on uniform matrices of 10M rows x 32 entries whose column count (= rows of the rhs, 128 bytes each) sweeps across
1 GiB of rhs: is there a cliff where the gathered operand outgrows what the address translation caches cover?"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402
from spmm_bound_probe import uniform_rows            # noqa: E402


def main():
    n, k = 10_000_000, 16
    dev = torch.device("cuda", 0)
    for cols in (2_000_000, 4_000_000, 6_000_000, 7_500_000, 8_388_608, 9_000_000, 10_000_000, 14_000_000, 20_000_000):
        indptr, indices, data = uniform_rows(n, cols, 32, dev, 13)
        a = DeviceCsMat.wrap_torch((n, cols), indptr, indices, data)
        rhs = gen.dense_vector(cols * k, seed=5, device=dev)
        out = torch.empty(n * k, dtype=torch.float64, device=dev)
        call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(a._h, C.c_void_p(rhs.data_ptr()), cols, k, k, C.c_void_p(out.data_ptr()), n, k, 0, None))
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for p, q in evs:
            p.record()
            call()
            q.record()
        torch.cuda.synchronize()
        ms = sum(p.elapsed_time(q) for p, q in evs) / len(evs)
        print(json.dumps({"cols": cols, "rhs_GiB": round(cols * k * 8 / 2**30, 3), "ms": round(ms, 3), "Grows_per_s": round(indices.numel() / ms / 1e6, 1),
                          "rhs_ptr": hex(rhs.data_ptr())}), flush=True)
        del a, indptr, indices, data, rhs, out


if __name__ == "__main__":
    main()
