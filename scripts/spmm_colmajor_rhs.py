#!/usr/bin/env python3
"""`&CsMat * &Array2` with a COLUMN-major rhs of 16 columns (result row-major, csmat.rs:2002-2016) on the bench matrix: with the rhs
gathered in place (a separate line per column and entry) and through the re-laid-out copy (option spmm_relayout)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sprs_amd                                      # noqa: E402
from sprs_amd import gen, prod                       # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec   # noqa: E402


def main():
    n, k = 10_000_000, 16
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    rhs_rm = gen.dense_vector(n * k, seed=5, device=dev).view(n, k)
    ref = None
    for name, col_major, relayout in (("row-major rhs", False, 0), ("column-major rhs, gathered in place", True, 2), ("column-major rhs, re-laid-out copy", True, 0)):
        sprs_amd.set_option("spmm_relayout", relayout)
        t = rhs_rm.t().contiguous() if col_major else rhs_rm.contiguous()      # (k, n) contiguous = column-major n x k
        m = prod.DeviceMat(n, k, DeviceVec.borrow(t.reshape(-1)), col_major=col_major)
        for _ in range(2):
            res = a * m
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            res = a * m
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        got = res.to_host()
        if ref is None:
            ref = got
        print(json.dumps({"case": name, "k": k, "ms": round(ms, 3), "max_abs_diff_vs_row_major": float(np.abs(got - ref).max())}), flush=True)
    sprs_amd.set_option("spmm_relayout", 0)


if __name__ == "__main__":
    main()
