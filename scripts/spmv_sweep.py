#!/usr/bin/env python3
"""A/B sweep of SpMV plan options on ONE generated matrix (default: the bench's R-MAT 10M): the matrix is
generated once, every configuration rebuilds the plan of the same handle and is timed with HIP events.
One JSON line per configuration on stdout.  Not the driver's bench: bench.py stays the graded harness.

  python scripts/spmv_sweep.py [--workload rmat10m] [--steps 20] CONFIG [CONFIG ...]
  CONFIG = name:opt=val,opt=val   (options of sprs_hip_set_option; unlisted ones are reset to defaults)

The first configuration's result is checked against the CPU oracle (tests' checker, --oracle), the
others against the first one on the device."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULTS = {}   # option -> library default, read from the library for every option the configurations name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rmat10m")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--idx-bytes", type=int, default=8)
    ap.add_argument("--oracle", action="store_true", help="check the first configuration against the CPU oracle")
    ap.add_argument("--repeat", type=int, default=1, help="run the whole list of configurations this many times, in rotation (run-to-run "
                                                          "drift of a box is several per cent: compare medians)")
    ap.add_argument("--row-block", default="", help="g/G: only the g-th of G cost-balanced row blocks (what rank g of a G-GPU run multiplies)")
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import sprs_amd
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    idt = torch.int64 if args.idx_bytes == 8 else torch.int32
    wl = args.workload
    if wl == "rmat10m":
        n, k = 10_000_000, 32
    elif wl == "rmat1m":
        n, k = 1_000_000, 16
    elif wl.startswith("rmat:"):
        _, n, k = wl.split(":")
        n, k = int(n), float(k)
    else:
        sys.exit("unknown workload")
    indptr, indices, data = gen.rmat_csr(n, k, device=dev, idx_dtype=idt, ptr_dtype=idt)
    rows = n
    if args.row_block:
        g, G = (int(v) for v in args.row_block.split("/"))
        cuts = gen.balanced_row_blocks(indptr, G, row_weight=8.0)
        r0, r1 = cuts[g], cuts[g + 1]
        lo, hi = int(indptr[r0]), int(indptr[r1])
        indptr = (indptr[r0:r1 + 1] - indptr[r0]).contiguous()
        indices, data = indices[lo:hi].clone(), data[lo:hi].clone()
        rows = r1 - r0
    nnz = indices.numel()
    x = gen.dense_vector(n, seed=3, device=dev)
    y = torch.zeros(rows, dtype=torch.float64, device=dev)
    a = DeviceCsMat.wrap_torch((rows, n), indptr, indices, data)
    xv, yv = DeviceVec.borrow(x), DeviceVec.borrow(y)
    stream = torch.cuda.current_stream()
    alg = nnz * (8 + args.idx_bytes) + (rows + 1) * args.idx_bytes + (n + rows) * 8
    first = None
    for spec in args.configs:
        for kv in filter(None, spec.partition(":")[2].split(",")):
            DEFAULTS.setdefault(kv.split("=")[0], sprs_amd.get_option(kv.split("=")[0]))
    for spec in list(args.configs) * args.repeat:
        name, _, rest = spec.partition(":")
        opts = dict(DEFAULTS)
        for kv in filter(None, rest.split(",")):
            key, val = kv.split("=")
            opts[key] = int(val)
        for key, val in opts.items():
            sprs_amd.set_option(key, val)
        a.refresh()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prod.csmat_mul_vec(a, xv, out=yv, stream=stream)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
        for _ in range(args.warmup):
            prod.csmat_mul_vec(a, xv, out=yv, stream=stream)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for s in range(args.steps):
            ev[s][0].record(stream)
            prod.csmat_mul_vec(a, xv, out=yv, stream=stream)
            ev[s][1].record(stream)
        torch.cuda.synchronize()
        ms = np.array([p.elapsed_time(q) for p, q in ev])
        kind, pbytes = a.spmv_plan_info()
        rec = {"config": name, "opts": {k: v for k, v in opts.items() if v != DEFAULTS[k]}, "ms_avg": round(float(ms.mean()), 4),
               "ms_min": round(float(ms.min()), 4), "gflops": round(2 * nnz / ms.mean() / 1e6, 1),
               "frac_of_8TBs": round(alg / (ms.mean() * 1e-3) / 8e12, 4), "plan_kind": kind, "plan_MB": round(pbytes / 1e6, 1),
               "first_call_ms": round(build_ms, 1)}
        if first is None:
            first = y.clone()
            if args.oracle:
                from oracle import oracle   # checker only
                npdt = np.uint64 if args.idx_bytes == 8 else np.uint32
                yh = np.zeros(rows)
                oracle.mul_acc_mat_vec_csr((rows, n), indptr.cpu().numpy().view(npdt), indices.cpu().numpy().view(npdt),
                                           data.cpu().numpy(), x.cpu().numpy(), yh)
                yg = y.cpu().numpy()
                den = np.maximum(np.abs(yh), np.abs(yg))
                rec["max_rel_err_vs_oracle"] = float(np.max(np.where(den > 0, np.abs(yg - yh) / np.where(den > 0, den, 1), 0)))
        else:
            den = torch.maximum(first.abs(), y.abs())
            rel = torch.where(den > 0, (y - first).abs() / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(den))
            rec["max_rel_diff_vs_first"] = float(rel.max().item())
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
