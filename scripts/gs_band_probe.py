#!/usr/bin/env python3
"""Step time of the Gauss-Seidel band schedule: a 5-point-like band matrix of `bands` bands (64 S rows each; the first row of
every chain holds the diagonal only, like a grid's border) swept in band mode and in level order.
usage: gs_band_probe.py [S=4096] [bands ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sprs_amd
from sprs_amd.device import DeviceCsMat, DeviceVec
from sprs_amd.linalg import gauss_seidel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if os.environ.get("GS_DEBUG"):
    sprs_amd.set_option("gauss_seidel_debug", int(os.environ["GS_DEBUG"]))
dev = torch.device("cuda", 0)
for bands in [int(v) for v in sys.argv[2:]] or [1, 2, 8, 64]:
    n = bands * 64 * S
    r = torch.arange(n, device=dev, dtype=torch.int64)
    border = (r % S == 0)
    offs = torch.tensor([-S, -1, 0, 1, S], device=dev, dtype=torch.int64)
    vals = torch.tensor([1.0, 1.0, -4.0, 1.0, 1.0], device=dev, dtype=torch.float64)
    cols = r[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < n) & (~border[:, None] | (offs[None, :] == 0))
    counts = ok.sum(1)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=indptr[1:])
    indices = cols[ok].contiguous()
    data = vals[None, :].expand(n, 5)[ok].contiguous()
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    rhs = DeviceVec.borrow(torch.ones(n, dtype=torch.float64, device=dev))
    out = {"S": S, "bands": bands, "rows": n}
    for mode, opt in (("band", S), ("level", 1)):
        sprs_amd.set_option("gauss_seidel_chain", opt)
        x = DeviceVec.borrow(torch.zeros(n, dtype=torch.float64, device=dev))
        gauss_seidel(a, x, rhs, 1, -1.0)
        torch.cuda.synchronize()
        ts = []
        for k in (2, 6):
            x = DeviceVec.borrow(torch.zeros(n, dtype=torch.float64, device=dev))
            t = time.perf_counter(); gauss_seidel(a, x, rhs, k, -1.0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out[mode + "_ms_per_sweep"] = round((ts[1] - ts[0]) / 4 * 1e3, 3)
        out[mode + "_x_sum"] = float(x.to_host().sum()) if n <= (1 << 22) else None
    out["us_per_step_band"] = round(out["band_ms_per_sweep"] * 1e3 / (S + 63 + (bands - 1) * 80), 3)
    sprs_amd.set_option("gauss_seidel_chain", 0)
    print(json.dumps(out), flush=True)
