#!/usr/bin/env python3
"""Probe for an SpMM with reuse: would a FIRST pass over the entries of the H most popular columns —
gathered from a compact, L2-sized copy of their rhs rows — plus a second pass over the rest beat one pass over everything?
Both parts are built here, in torch, as ordinary CSR matrices (the hot part with its columns renumbered 0 .. H - 1 by popularity)
and multiplied by the library's existing SpMM entry: hot part in the operator form, the rest accumulating.  Nothing in the library
changes; the probe only says whether the split is worth building.
usage: spmm_two_pass_probe.py [n] [nnz_per_row] [k] [H ...]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402


def timed(call, reps=8):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nzr = float(sys.argv[2]) if len(sys.argv) > 2 else 32
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    hs = [int(v) for v in sys.argv[4:]] or [8192, 16384, 32768, 65536]
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, nzr, device=dev)
    nnz = indices.numel()
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    rhs = gen.dense_vector(n * k, seed=5, device=dev)
    out = torch.empty(n * k, dtype=torch.float64, device=dev)
    spmm = lambda h, r, rrows, o, acc: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(h._h, C.c_void_p(r.data_ptr()), rrows, k, k, C.c_void_p(o.data_ptr()), n, k, acc, None))
    one = timed(lambda: spmm(a, rhs, n, out, 0))
    ref = out.clone()
    print(json.dumps({"what": "one pass (the library as it is)", "k": k, "ms": round(one, 4)}), flush=True)
    cnt = torch.bincount(indices, minlength=n)
    order = torch.argsort(cnt, descending=True, stable=True)
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(n, device=dev)
    rows_of = torch.repeat_interleave(torch.arange(n, device=dev), indptr[1:] - indptr[:-1])
    lab = rank[indices]
    for H in hs:
        hot = lab < H
        def part(mask, cols):
            ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            ip[1:] = torch.cumsum(torch.bincount(rows_of[mask], minlength=n), 0)
            return ip, cols[mask].contiguous(), data[mask].contiguous()
        ip_h, ix_h, dt_h = part(hot, lab)                     # columns renumbered: row order inside a row is by ORIGINAL column, which the kernel does not need sorted
        ip_c, ix_c, dt_c = part(~hot, indices)
        ah = DeviceCsMat.wrap_torch((n, H), ip_h, ix_h, dt_h)
        ac = DeviceCsMat.wrap_torch((n, n), ip_c, ix_c, dt_c)
        rhs_hot = rhs.view(n, k)[order[:H]].contiguous().view(-1)        # the compact copy: H x k doubles
        t_copy = timed(lambda: rhs.view(n, k)[order[:H]].contiguous())
        t_hot = timed(lambda: spmm(ah, rhs_hot, H, out, 0))
        t_cold = timed(lambda: spmm(ac, rhs, n, out, 1))
        spmm(ah, rhs_hot, H, out, 0)
        spmm(ac, rhs, n, out, 1)
        torch.cuda.synchronize()
        err = float(((out - ref).abs() / ref.abs().clamp_min(1e-300)).max())
        print(json.dumps({"what": "two passes", "k": k, "H": H, "hot_MB": round(H * k * 8 / 1e6, 2), "hot_share_of_entries": round(float(hot.float().mean()), 4),
                          "hot_pass_ms": round(t_hot, 4), "cold_pass_ms": round(t_cold, 4), "rhs_copy_ms": round(t_copy, 4),
                          "sum_ms": round(t_hot + t_cold + t_copy, 4), "one_pass_ms": round(one, 4), "max_rel_diff_vs_one_pass": err}), flush=True)
        del ah, ac, ip_h, ix_h, dt_h, ip_c, ix_c, dt_c, rhs_hot
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
