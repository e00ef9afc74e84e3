#!/usr/bin/env python3
"""SpMV of ONE row block of a G-way cut of the bench matrix (what a rank of `bench.py --gpus G` multiplies), for kernel traces:
usage: [SPRS_OPTS="name=value ..."] block_spmv.py G g [reps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import gen, prod                       # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec   # noqa: E402


def main():
    G, g = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    n = 10_000_000
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    x = gen.dense_vector(n, device=dev)
    cuts = gen.balanced_row_blocks(indptr, G, row_weight=8.0)
    r0, r1 = cuts[g], cuts[g + 1]
    lo, hi = int(indptr[r0]), int(indptr[r1])
    ip = (indptr[r0:r1 + 1] - indptr[r0]).contiguous()
    a = DeviceCsMat.wrap_torch((r1 - r0, n), ip, indices[lo:hi].clone(), data[lo:hi].clone())
    y = torch.empty(r1 - r0, dtype=torch.float64, device=dev)
    xs, ys = DeviceVec.borrow(x), DeviceVec.borrow(y)
    opts = {}
    for opt in (os.environ.get("SPRS_OPTS") or "").split():
        import sprs_amd
        key, val = opt.split("=")
        sprs_amd.set_option(key, int(val))
        opts[key] = int(val)
    for _ in range(3):
        prod.csmat_mul_vec(a, xs, out=ys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        prod.csmat_mul_vec(a, xs, out=ys)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps({"G": G, "block": g, "rows": r1 - r0, "nnz": hi - lo, "opts": opts, "ms": round(ms, 4), "plan": list(a.spmv_plan_info())}))


if __name__ == "__main__":
    main()
