#!/usr/bin/env python3
"""BASELINE config 5 (C = A*A, R-MAT 1M x 1M ~8 nnz/row), NOTHING ELSE: generation + <products> products.  The profiling
target of the SpGEMM counter passes (rocprofv3 --pmc serialises and instruments every dispatch: anything beside the
product — parity checks over 3.3e9 entries, a second product — only makes the pass miss its time budget).
usage: spgemm_one.py [products=1] [idx_bytes=8]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import gen, smmp                      # noqa: E402
from sprs_amd.device import DeviceCsMat              # noqa: E402


def main():
    products = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    idx_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    n = 1_000_000
    idt = torch.int64 if idx_bytes == 8 else torch.int32
    indptr, indices, data = gen.rmat_csr(n, 8, device=dev, idx_dtype=idt, ptr_dtype=torch.int64, oversample=1.0)
    if os.environ.get("SPGEMM_PERMUTE"):                      # P A P^T with a random P (0: identity, the control): tests/spgemm_bench.py
        g = torch.Generator(device=dev)
        g.manual_seed(int(os.environ["SPGEMM_PERMUTE"]))
        perm = torch.randperm(n, device=dev, generator=g)
        if os.environ["SPGEMM_PERMUTE"] == "0":
            perm = torch.arange(n, device=dev)
        rows_of = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
        key, order = torch.sort((perm[rows_of] << 32) | perm[indices.long()])
        new_rows = key >> 32
        indices = (key & 0xFFFFFFFF).to(idt)
        data = data[order]
        indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        indptr[1:] = torch.cumsum(torch.bincount(new_rows, minlength=n), 0)
        del perm, rows_of, key, order, new_rows
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    torch.cuda.synchronize()
    print("generated: nnz(A) = %d" % indices.numel(), file=sys.stderr, flush=True)
    times, c = [], None
    for _ in range(products):
        c = None
        t0 = time.perf_counter()
        c = smmp.mul_csr_csr(a, a)
        torch.cuda.synchronize()
        times.append(round(time.perf_counter() - t0, 4))
    print(json.dumps({"products": products, "seconds": times, "nnz_c": int(c.nnz()), "idx_bytes": idx_bytes}))


if __name__ == "__main__":
    main()
