#!/usr/bin/env python3
"""Scaling MODEL from one GPU (only one MI355X is reachable from the build environment): cut the
bench matrix into G nnz-balanced row blocks exactly as bench.py --gpus G does, time each block's
SpMV on this GPU, and report max-over-blocks (= the compute part of a G-GPU step) plus a modelled
direct all-gather-v time over xGMI (7 links x ~153 GB/s per GPU, each rank pushes its y block to
its G-1 peers concurrently).  Clearly a model, never a measurement of G GPUs.
usage: virtual_ranks.py [n] [nnz_per_row]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import gen, prod                    # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec   # noqa: E402

LINK_GBS = 153.0
RW = float(os.environ.get("ROW_WEIGHT", "5"))


def time_spmv(a, x, y, reps=20):
    xs, ys = DeviceVec.borrow(x), DeviceVec.borrow(y)
    for _ in range(3):
        prod.csmat_mul_vec(a, xs, out=ys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        prod.csmat_mul_vec(a, xs, out=ys)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    k = float(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, k, device=dev)
    x = gen.dense_vector(n, device=dev)
    nnz = indices.numel()
    out = {"n": n, "nnz": nnz, "row_weight": RW, "model": "per-block kernel time measured on ONE MI355X; exchange modelled"}
    # ROW_WEIGHTS="8,16,32": the cost function nnz + w * rows of the partition swept (choose it to minimise
    # max(compute) + max(y block) / one xGMI link) — one JSON line per weight, G = 4 and 8 only
    if os.environ.get("ROW_WEIGHTS"):
        for w in [float(v) for v in os.environ["ROW_WEIGHTS"].split(",")]:
            rec = {"row_weight": w}
            for G in (4, 8):
                cuts = gen.balanced_row_blocks(indptr, G, row_weight=w)
                times, rows, nz = [], [], []
                for g in range(G):
                    r0, r1 = cuts[g], cuts[g + 1]
                    lo, hi = int(indptr[r0]), int(indptr[r1])
                    a = DeviceCsMat.wrap_torch((r1 - r0, n), (indptr[r0:r1 + 1] - indptr[r0]).contiguous(), indices[lo:hi].clone(), data[lo:hi].clone())
                    y = torch.empty(r1 - r0, dtype=torch.float64, device=dev)
                    times.append(time_spmv(a, x, y))
                    rows.append(r1 - r0)
                    nz.append(hi - lo)
                    del a, y
                gather_s = max(rows) * 8 / (LINK_GBS * 1e9)
                rec["G=%d" % G] = {"compute_ms": [round(t * 1e3, 4) for t in times], "rows": rows, "nnz": nz,
                                   "modelled_allgather_ms": round(gather_s * 1e3, 4), "modelled_step_ms": round((max(times) + gather_s) * 1e3, 4),
                                   "modelled_gflops": round(2 * nnz / (max(times) + gather_s) / 1e9, 1)}
            print(json.dumps(rec), flush=True)
        return
    for G in (1, 2, 4, 8):
        cuts = gen.balanced_row_blocks(indptr, G, row_weight=RW)
        times, rows, subs = [], [], []
        for g in range(G):
            r0, r1 = cuts[g], cuts[g + 1]
            lo, hi = int(indptr[r0]), int(indptr[r1])
            ip = (indptr[r0:r1 + 1] - indptr[r0]).contiguous()
            ix, dt = indices[lo:hi].clone(), data[lo:hi].clone()
            a = DeviceCsMat.wrap_torch((r1 - r0, n), ip, ix, dt)
            y = torch.empty(r1 - r0, dtype=torch.float64, device=dev)
            times.append(time_spmv(a, x, y))
            rows.append(r1 - r0)
            # the same block as TWO sub-blocks of equal cost (nnz + 8 per row: the cut of dist.hip, nsub = 2), each with its own plan
            if G > 1:
                cost = ip.double() + 8.0 * torch.arange(ip.numel(), device=dev, dtype=torch.float64)
                c = int(torch.searchsorted(cost, cost[-1] / 2).item())
                c = min(max(c, 1), (r1 - r0) - 1)
                ts = []
                for (s0, s1) in ((0, c), (c, r1 - r0)):
                    ip_s = (ip[s0:s1 + 1] - ip[s0]).contiguous()
                    l0, l1 = int(ip[s0]), int(ip[s1])
                    a_s = DeviceCsMat.wrap_torch((s1 - s0, n), ip_s, ix[l0:l1].clone(), dt[l0:l1].clone())
                    ts.append(time_spmv(a_s, x, y[s0:s1]))
                    del a_s
                subs.append(ts)
            del a, ip, ix, dt, y
        biggest = max(rows) * 8
        # direct exchange: the rank with the biggest y block pushes it to G-1 peers over G-1 links
        gather_s = 0.0 if G == 1 else biggest / (LINK_GBS * 1e9)
        step = max(times) + gather_s
        out["G=%d" % G] = {"compute_ms_max": round(max(times) * 1e3, 4), "compute_ms_min": round(min(times) * 1e3, 4),
                           "rows_per_block": rows, "modelled_allgather_ms": round(gather_s * 1e3, 4),
                           "modelled_step_ms": round(step * 1e3, 4),
                           "modelled_gflops": round(2 * nnz / step / 1e9, 1)}
        if subs:
            # nsub = 2: sub-block 0, then its exchange beside sub-block 1, then the exchange of sub-block 1 (half the bytes each)
            step2 = max(t[0] + max(t[1], gather_s / 2) + gather_s / 2 for t in subs)
            out["G=%d" % G].update({"nsub2_compute_ms_per_block": [[round(v * 1e3, 4) for v in t] for t in subs],
                                    "nsub2_modelled_step_ms": round(step2 * 1e3, 4),
                                    "nsub2_modelled_gflops": round(2 * nnz / step2 / 1e9, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
