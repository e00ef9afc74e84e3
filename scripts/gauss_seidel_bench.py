#!/usr/bin/env python3
"""Gauss-Seidel sweep (twin of gauss_seidel() of the reference's heat example, heat.rs:103-139) on the heat system of a
G x G grid: time per sweep on the device (sweep kernel + the residual SpMV + the convergence scalar, as the reference's loop
has them), the one-time plan (level order on the device, SpMV plan), the CPU oracle's time per sweep beside it, and parity:
the iterate after K sweeps bit for bit against the oracle (at the full size: the oracle sweeps 1.7e7 rows in ~0.4 s).
usage: gauss_seidel_bench.py [G] [K] [blocks[:naps[:xcd]] ...]     (defaults 4096 3 and the library's defaults; blocks = workgroups
of the sweep kernel, naps = longest pause of a waiting wave)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sprs_amd                                               # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec           # noqa: E402
from sprs_amd.linalg import gauss_seidel                      # noqa: E402
from oracle import oracle                                     # noqa: E402  (the checker and the CPU baseline, nothing else)


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    blocks = [tuple(int(t) for t in (v + ":0:0").split(":")[:3]) for v in sys.argv[3:]] or [(0, 0, 0)]
    if os.environ.get("GS_CHAIN"):                       # 1: level order only; S > 1: the band schedule with this stride (default auto)
        sprs_amd.set_option("gauss_seidel_chain", int(os.environ["GS_CHAIN"]))
    if os.environ.get("GS_BANDED"):
        # experiment: the same number of rows, entries per row and levels as the heat system, but the rows of a level are
        # CONTIGUOUS (row i reads rows i - W and i - W + 1): hand-offs of neighbouring lanes share cache lines
        w = int(os.environ["GS_BANDED"])
        n = g * g
        i = np.arange(n, dtype=np.int64)
        cols = np.stack([i - w, i - w + 1, i, i + 1, i + w], axis=1)
        vals = np.broadcast_to(np.array([1.0, 1.0, -4.5, 1.0, 1.0]), cols.shape)
        keep = (cols >= 0) & (cols < n)
        keep[:, 1] &= cols[:, 1] < i                           # (i - W + 1 < i unless W = 1)
        lens = keep.sum(axis=1)
        ip = np.zeros(n + 1, dtype=np.uint64)
        ip[1:] = np.cumsum(lens)
        ix = cols[keep].astype(np.uint64)
        dt = np.ascontiguousarray(vals[keep])
        shape = (n, n)
        rhs = np.random.default_rng(1).standard_normal(n)
        del cols, vals, keep, i
    elif g > 0:
        shape, ip, ix, dt = oracle.grid_laplacian(g, g)
        n = g * g
        i, j = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
        border = (i == 0) | (i == g - 1) | (j == 0) | (j == g - 1)
        rhs = np.where(border, (i + j).astype(np.float64), 0.0).reshape(-1)
        del i, j, border
    else:
        # G < 0: a random non-symmetric, strictly diagonally dominant system of -G rows, 8 off-diagonal entries per row
        # (few, wide levels: the sweep is bound by throughput there, not by the chain of levels)
        import scipy.sparse as sp
        n = -g
        rng = np.random.default_rng(1)
        r = np.repeat(np.arange(n), 8)
        c = rng.integers(0, n, size=8 * n)
        m = sp.coo_matrix((rng.standard_normal(8 * n), (r, c)), shape=(n, n)).tocsr()
        m = (m + sp.diags(np.abs(m).sum(axis=1).A1 + 1.0)).tocsr()
        m.sort_indices()
        shape, ip, ix, dt = (n, n), m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data
        rhs = rng.standard_normal(n)
        del m, r, c
    x0 = np.zeros(n)
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    d_rhs = DeviceVec.from_host(rhs)
    t0 = time.perf_counter()
    x = DeviceVec.from_host(x0)
    r = gauss_seidel(a, x, d_rhs, 1, -1.0)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    x_ref, info = oracle.gauss_seidel(shape, ip, ix, dt, x0, rhs, k, -1.0)
    cpu = (time.perf_counter() - t0) / max(k, 1)
    alg = ix.size * (8 + ix.itemsize) + (n + 1) * ip.itemsize + 4 * n * 8      # matrix once, x old + new, rhs, order (4 B) ~ per sweep
    for b, naps, xcd in blocks:
        sprs_amd.set_option("gauss_seidel_blocks", b)
        sprs_amd.set_option("gauss_seidel_naps", naps)
        sprs_amd.set_option("gauss_seidel_xcd", xcd)
        x = DeviceVec.from_host(x0)
        gauss_seidel(a, x, d_rhs, 1, -1.0)                     # warm
        times = {}
        for kk in (k, 4 * k):
            x = DeviceVec.from_host(x0)
            t0 = time.perf_counter()
            res = gauss_seidel(a, x, d_rhs, kk, -1.0)
            times[kk] = time.perf_counter() - t0
            if kk == k:
                got = x.to_host()
        per = (times[4 * k] - times[k]) / (3 * k)
        same = bool(np.array_equal(got, x_ref))
        print(json.dumps({"grid": g, "rows": n, "nnz": int(ix.size), "levels": res.levels, "workgroups": b or "default", "naps": naps or "default", "xcd_mode": xcd,
                          "ms_per_sweep_with_residual": round(per * 1e3, 3), "us_per_level": round(per * 1e6 / res.levels, 3),
                          "first_call_s_plans_included": round(first, 3), "oracle_ms_per_sweep_with_residual": round(cpu * 1e3, 1),
                          "speedup_vs_one_core": round(cpu / per, 1), "streamed_GBs": round(alg / per / 1e9, 1),
                          "iterate_after_%d_sweeps_bit_identical" % k: same}), flush=True)


if __name__ == "__main__":
    main()
