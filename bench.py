#!/usr/bin/env python3
"""bench.py — CSR SpMV throughput of the HIP path on BASELINE.json's headline
configuration: R-MAT 10M x 10M, ~32 nnz/row, f64, sprs-default usize (64-bit)
indices and indptr.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one y = A*x over the whole matrix (for N > 1: the local row-block
multiply plus the all-gather-v of y over xGMI/RCCL).  The matrix, x and y are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Workload selection (for parity-sized runs, never for the headline):
  --workload rmat10m (default) | rmat1m | laplace4096 | rmat:<n>:<nnz_per_row>
  --workload spgemm_uniform   the reference's own bench shape (sprs-benches/src/main.rs:148-163): uniform density, 2.5M x 2.5M,
                        4 entries per row, oracle at Fixed(1) and Automatic, whole-product parity; also an object of the default line
  --workload spgemm5    BASELINE config 5: C = A * A, R-MAT 1M x 1M ~8 nnz/row (smmp::mul_csr_csr twin); a step is one
                        whole product; prints its own JSON line (seconds per product, compulsory-bytes roofline,
                        CPU baseline = the oracle's mul_csr_csr with sprs' chunking at T = 1 and Automatic on sampled
                        row blocks, row-block parity).  The default SpMV line carries a short "spgemm5" object too
                        (--no-secondary skips it).
  --idx-bytes 8 (default, sprs usize) | 4
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(rows, cols, nnz, idx_bytes, iptr_bytes, accumulate=False):
    """nnz*(8+S_I) + (rows+1)*S_P + cols*8 [x once] + rows*8 [y] (+rows*8 if accumulating)."""
    return nnz * (8 + idx_bytes) + (rows + 1) * iptr_bytes + cols * 8 + rows * 8 * (2 if accumulate else 1)


# what a workload's launches can reach: its kernels, the headers they include, the shared scan / sort, the option table
CSRC_OF = {
    "spmv": ("common.hpp", "scan.hpp", "scan.hip", "spmv_shared.hpp", "spmv.hip", "spmv_band.hip", "spmv_band_kernels.hpp"),
    "spgemm": ("common.hpp", "lanes.hpp", "scan.hpp", "scan.hip", "sort.hip", "spgemm.hip"),
}


def csrc_sha16(part=None):
    """hash of kernel sources: ties a committed PMC measurement to the code it was taken on.  part = "spmv" / "spgemm":
    the translation units and headers that workload runs through (a change to another kernel family leaves the
    measurement valid); None: every *.hip / *.hpp of sprs_amd/csrc."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sprs_amd", "csrc")
    files = [os.path.join(d, f) for f in CSRC_OF[part]] if part else glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp"))
    for f in sorted(files):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def spgemm5(dev, idx_bytes, steps, warmup, check_rows, cpu_blocks, cpu_block_rows, with_cpu=True):
    """BASELINE config 5: C = A * A for R-MAT 1M x 1M, ~8 nnz/row (7.78e6 entries; nnz(C) = 3.3e9, 5.5e9 products).
    Returns the JSON object of the workload.  SURVEY 8(d): compulsory bytes = (nnzA + nnzB + nnzC)(V + S_I) + 3 (n + 1) S_P."""
    import sprs_amd
    from sprs_amd import gen, smmp
    from sprs_amd.device import DeviceCsMat
    n, k = 1_000_000, 8
    idt = torch.int64 if idx_bytes == 8 else torch.int32
    indptr, indices, data = gen.rmat_csr(n, k, device=dev, idx_dtype=idt, ptr_dtype=torch.int64, oversample=1.0)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    rl = (indptr[1:] - indptr[:-1]).to(torch.float64)
    csz = torch.zeros(indices.numel() + 1, dtype=torch.float64, device=dev)
    csz[1:] = torch.cumsum(rl[indices.long()], 0)          # multiply-adds of the entries before each position
    per_row_products = csz[indptr.long()]                  # ... of the rows before each row (n + 1 values)
    del csz
    products = float(per_row_products[-1].item())
    stream = torch.cuda.current_stream()
    c = None
    for _ in range(warmup):
        c = None
        c = smmp.mul_csr_csr(a, a)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s_ in range(steps):
        c = None                                   # the previous result goes back to the library's block pool
        ev[s_][0].record(stream)
        c = smmp.mul_csr_csr(a, a)                 # symbolic + prefix sum + numeric, result resident in HBM
        ev[s_][1].record(stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = [p.elapsed_time(q) for p, q in ev]
    nnz_a, nnz_c = int(indices.numel()), int(c.nnz())
    comp = (2 * nnz_a + nnz_c) * (8 + idx_bytes) + 3 * (n + 1) * 8
    sec = float(np.mean(ms)) * 1e-3
    out = {
        "metric": "CSR x CSR SpGEMM seconds per product (A*A, R-MAT 1M ~8/row)", "value": round(sec, 5), "unit": "s",
        "higher_is_better": False, "steps": steps, "warmup": warmup, "ms_per_step": round(wall * 1e3, 3), "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: C = A*A, R-MAT 1M x 1M ~8 nnz/row (seed 1, no oversampling)", "rows": n, "nnz_a": nnz_a,
                   "nnz_c": nnz_c, "products": products, "index_bytes": idx_bytes, "indptr_bytes": 8},
        "gflops": round(2 * products / sec / 1e9, 2),
        "roofline": {"bound": "hbm", "kernel": "all kernels of one sprs_hip_spgemm_f64 call (row_work, task lists, symbolic, scans, numeric)",
                     "achieved": round(comp / sec / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(comp / sec / 1e9 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_launch": comp, "no_reuse_upper_bound_bytes": int(products * (8 + idx_bytes)),
                     "kernel_ms_avg": round(float(np.mean(ms)), 3), "kernel_ms_min": round(float(np.min(ms)), 3), "traffic": None},
    }
    try:        # PMC traffic of the same product on exactly these kernel sources (scripts/spgemm_traffic.py), else null
        sha = csrc_sha16("spgemm")
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if e["workload"] == "spgemm5" and e["index_bytes"] == idx_bytes and e.get("csrc_sha16") == sha:
                    out["roofline"]["traffic"] = e["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    # ---- parity and CPU baseline on row blocks: the oracle is the checker and the timed CPU port, never the product -------
    from oracle import oracle
    npi = np.uint64 if idx_bytes == 8 else np.uint32
    ip_h = indptr.cpu().numpy().view(np.uint64)
    ix_h = indices.cpu().numpy().view(npi)
    dt_h = data.cpu().numpy()
    ok, worst, checked = True, 0.0, 0
    for r0 in sorted(set([0, n // 3, max(0, n - check_rows)])):
        r1 = min(n, r0 + check_rows)
        s0, e0 = int(ip_h[r0]), int(ip_h[r1])
        _, rip, rix, rdt = oracle.mul_csr_csr((r1 - r0, n), ip_h[r0:r1 + 1], ix_h[s0:e0], dt_h[s0:e0], (n, n), ip_h, ix_h, dt_h, threads=0)
        gip, gix, gdt = c.slice_outer_to_host(r0, r1)
        gip = gip - gip[0]
        ok &= bool(np.array_equal(gip, rip)) and bool(np.array_equal(gix, rix))
        if ok and rdt.size:
            worst = max(worst, float(np.max(np.abs(gdt - rdt) / np.maximum(np.abs(rdt), 1e-300))))
        checked += int(rix.size)
    out["parity"] = {"rows_checked": 3 * check_rows, "entries_checked": checked, "structure_bit_exact": ok, "max_rel_err": worst,
                     "tolerance": 1e-10, "ok": bool(ok and worst <= 1e-10)}
    if with_cpu:
        # evenly spaced row blocks; the whole product is extrapolated by the ratio of multiply-adds (products)
        pr_h = per_row_products.cpu().numpy()
        starts = [int(i * (n - cpu_block_rows) / max(1, cpu_blocks - 1)) for i in range(cpu_blocks)]
        p_sample, t1, tauto, tphys, used_auto, phys_threads = 0.0, 0.0, 0.0, 0.0, 0, 1
        for r0 in starts:
            r1 = r0 + cpu_block_rows
            s0, e0 = int(ip_h[r0]), int(ip_h[r1])
            blk = ((r1 - r0, n), ip_h[r0:r1 + 1], ix_h[s0:e0], dt_h[s0:e0], (n, n), ip_h, ix_h, dt_h)
            t = time.perf_counter()
            oracle.mul_csr_csr(*blk, threads=1)
            t1 += time.perf_counter() - t
            t = time.perf_counter()
            res = oracle.mul_csr_csr(*blk, threads=0, return_threads=True)
            dt_auto = time.perf_counter() - t
            tauto += dt_auto
            used_auto = max(used_auto, res[-1])
            pt = oracle.automatic_physical_threads(e0 - s0, ix_h.size)
            phys_threads = max(phys_threads, pt)
            if pt != res[-1]:                               # ThreadingStrategy::AutomaticPhysical resolves to another thread count here
                t = time.perf_counter()
                oracle.mul_csr_csr(*blk, threads=pt)
                tphys += time.perf_counter() - t
            else:
                tphys += dt_auto                            # the same thread count: the same run
            p_sample += float(pr_h[r1] - pr_h[r0])
        scale = products / p_sample
        out["cpu_baseline"] = {
            "value": round(tauto * scale, 2), "unit": "s", "cores": int(used_auto), "kind": "port",
            "sample": "%d blocks of %d rows spread over the matrix (%.2f %% of the multiply-adds), each multiplied by the whole B with the C "
                      "restatement of smmp::mul_csr_csr and sprs' thread-count rule (ThreadingStrategy::Automatic: min(cores, "
                      "(nnzA + nnzB) / 8128), smmp.rs:210-227); extrapolated to the whole product by the ratio of multiply-adds; "
                      "rustc is not available here" % (cpu_blocks, cpu_block_rows, 100.0 / scale),
            "sample_seconds": round(tauto, 3), "single_thread_value": round(t1 * scale, 2), "single_thread_sample_seconds": round(t1, 3),
            "host_cores": oracle.num_procs(), "host_physical_cores": oracle.num_physical_cores(),
            # ThreadingStrategy::AutomaticPhysical (smmp.rs:26-31): the same rule on the physical cores
            "automatic_physical_value": round(tphys * scale, 2), "automatic_physical_cores": int(phys_threads),
        }
    # ---- the supported opt-out of the reference's ORDER of additions (option spgemm_ordered = 0): same products, added by the
    # waves of a large-row workgroup as they arrive.  Timed beside the default, never instead of it; compared entry by entry
    # with the ordered product (which is bit-identical to the oracle's) on the device.
    import ctypes as C
    from sprs_amd import _ffi
    sprs_amd.set_option("spgemm_ordered", 0)
    try:
        c2 = None
        for _ in range(max(1, min(warmup, 2))):
            c2 = None
            c2 = smmp.mul_csr_csr(a, a)
        torch.cuda.synchronize()
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for s_ in range(steps):
            c2 = None
            ev2[s_][0].record(stream)
            c2 = smmp.mul_csr_csr(a, a)
            ev2[s_][1].record(stream)
        torch.cuda.synchronize()
    finally:
        sprs_amd.set_option("spgemm_ordered", 1)
    ms2 = [p.elapsed_time(q) for p, q in ev2]
    ptrs = []
    for m in (c, c2):
        p_ip, p_ix, p_dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _ffi.check(_ffi.lib.sprs_hip_csmat_device_ptrs(m._h, C.byref(p_ip), C.byref(p_ix), C.byref(p_dt)))
        ptrs.append((p_ix.value, p_dt.value))
    blk = 1 << 26
    b0 = torch.empty(blk, dtype=torch.float64, device=dev)
    b1 = torch.empty(blk, dtype=torch.float64, device=dev)
    same_structure, worst = int(c2.nnz()) == nnz_c, 0.0
    for lo in range(0, nnz_c if same_structure else 0, blk):
        m = min(blk, nnz_c - lo)
        for buf, (pix, pdt) in ((b0, ptrs[0]), (b1, ptrs[1])):
            _ffi.check(_ffi.lib.sprs_hip_memcpy_d2d(C.c_void_p(buf.data_ptr()), C.c_void_p(pix + lo * idx_bytes), m * idx_bytes, None))
        torch.cuda.synchronize()
        same_structure = same_structure and bool(torch.equal(b0.view(torch.uint8)[:m * idx_bytes], b1.view(torch.uint8)[:m * idx_bytes]))
        for buf, (pix, pdt) in ((b0, ptrs[0]), (b1, ptrs[1])):
            _ffi.check(_ffi.lib.sprs_hip_memcpy_d2d(C.c_void_p(buf.data_ptr()), C.c_void_p(pdt + lo * 8), m * 8, None))
        torch.cuda.synchronize()
        worst = max(worst, float(((b0[:m] - b1[:m]).abs() / b0[:m].abs().clamp_min(1e-300)).max().item()))
    out["unordered_adds"] = {"option": "spgemm_ordered = 0", "seconds_per_product": round(float(np.mean(ms2)) * 1e-3, 5),
                             "kernel_ms_min": round(float(np.min(ms2)), 3), "indices_bit_identical_to_ordered": bool(same_structure),
                             "max_rel_diff_vs_ordered": worst, "tolerance": 1e-10, "ok": bool(same_structure and worst <= 1e-10)}
    del c2, b0, b1
    return out


def spmv_config(dev, wl, steps=30, warmup=4, with_oracle=True):
    """One SpMV configuration of BASELINE.json beside the headline (configs 2 and 3: `rmat1m`, `laplace4096`), timed like the
    headline (HIP events around every SpMV on the launch stream, matrix / x / y resident in HBM, handle prepared) — WARM
    (steps back to back: whatever fits stays in L2 / the 256 MiB Infinity Cache) and COLD (a 1 GiB read-modify-write
    between the steps, outside the events: matrix and x come from HBM) — with the WHOLE result vector checked against
    the oracle.  usize indices and indptr (the sprs default), like the headline."""
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    if wl == "rmat1m":
        n = 1_000_000
        indptr, indices, data = gen.rmat_csr(n, 16, device=dev)
        name = "BASELINE config 2: R-MAT 1M x 1M, ~16 nnz/row"
    elif wl == "laplace4096":
        n = 4096 * 4096
        indptr, indices, data = gen.grid_laplacian(4096, 4096, device=dev)
        name = "BASELINE config 3: 5-pt Laplacian 4096^2 grid"
    else:
        raise ValueError(wl)
    nnz = int(indices.numel())
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    x = gen.dense_vector(n, seed=3, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    xv, yv = DeviceVec.borrow(x), DeviceVec.borrow(y)
    stream = torch.cuda.current_stream()
    a.prepare(stream=stream)
    for _ in range(warmup):
        prod.csmat_mul_vec(a, xv, out=yv, stream=stream)
    torch.cuda.synchronize()
    alg = algorithmic_bytes(n, n, nnz, 8, 8)
    plan_kind, plan_bytes = a.spmv_plan_info()
    res = {"workload": name, "rows": n, "nnz": nnz, "index_bytes": 8, "algorithmic_bytes_per_launch": alg, "steps": steps,
           "plan": {1: "nnz tiles", 2: "xcd-sliced copy", 3: "banded copy (hot columns from LDS)"}.get(plan_kind, "none"),
           "plan_bytes": plan_bytes}
    flush = torch.empty(1 << 27, dtype=torch.float64, device=dev)       # 1 GiB
    for label, cold in (("warm", False), ("cold", True)):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for p_, q_ in ev:
            if cold:
                flush.add_(1.0)
            p_.record(stream)
            prod.csmat_mul_vec(a, xv, out=yv, stream=stream)
            q_.record(stream)
        torch.cuda.synchronize()
        ms = [p_.elapsed_time(q_) for p_, q_ in ev]
        avg = float(np.mean(ms))
        res[label] = {"kernel_ms_avg": round(avg, 5), "kernel_ms_min": round(float(np.min(ms)), 5),
                      "gflops": round(2.0 * nnz / (avg * 1e-3) / 1e9, 2), "achieved_GBs": round(alg / (avg * 1e-3) / 1e9, 1),
                      "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del flush
    if with_oracle:
        from oracle import oracle   # test infrastructure: the checker, never the product
        y_h = np.zeros(n)
        oracle.mul_acc_mat_vec_csr((n, n), indptr.cpu().numpy().view(np.uint64), indices.cpu().numpy().view(np.uint64),
                                   data.cpu().numpy(), x.cpu().numpy(), y_h)
        y_g = y.cpu().numpy()
        den = np.maximum(np.abs(y_h), np.abs(y_g))
        rel = np.where(den > 0, np.abs(y_g - y_h) / np.where(den > 0, den, 1.0), 0.0)
        if wl == "laplace4096":
            # the stencil rows cancel (1 + 1 - 4 + 1 + 1): the bound is componentwise against (|A| |x|)_i, as the GPU tests use
            ax = np.zeros(n)
            oracle.mul_acc_mat_vec_csr((n, n), indptr.cpu().numpy().view(np.uint64), indices.cpu().numpy().view(np.uint64),
                                       np.abs(data.cpu().numpy()), np.abs(x.cpu().numpy()), ax)
            rel = np.where(ax > 0, np.abs(y_g - y_h) / np.where(ax > 0, ax, 1.0), np.abs(y_g - y_h))
        res["parity"] = {"rows_checked": n, "max_rel_err_vs_oracle": float(rel.max()), "tolerance": 1e-10,
                         "bound": "|dy_i| <= tol (|A||x|)_i" if wl == "laplace4096" else "|dy_i| <= tol max(|y_i|, |ref_i|)",
                         "ok": bool(rel.max() <= 1e-10)}
    return res


def spgemm_uniform(dev, n=2_500_000, nnz_over_rows=4, steps=3, with_cpu=True):
    """The largest product of the reference's own bench (sprs-benches/src/main.rs:148-163, 178-186, 211-260): m1 (n x n) *
    m2 (n x n), both from the uniform generator at density nnz_over_rows / n (sprs-rand `rand_csr`, here gen.uniform_csr),
    timed at ThreadingStrategy::Fixed(1) and ::Automatic on the CPU (the C restatement of smmp::mul_csr_csr; the reference
    also times 2 and 4 threads).  Every row of the product has ~16 multiply-adds: the whole product runs through the
    lane-group kernel of the rows of at most 64 products (micro_rows_kernel).  The WHOLE product is compared
    with the oracle's, structure and value bits."""
    from sprs_amd import gen, smmp
    from sprs_amd.device import DeviceCsMat
    dens = float(nnz_over_rows) / n
    a_ip, a_ix, a_dt = gen.uniform_csr((n, n), dens, seed=11, value_seed=12, device=dev)
    b_ip, b_ix, b_dt = gen.uniform_csr((n, n), dens, seed=21, value_seed=22, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), a_ip, a_ix, a_dt)
    b = DeviceCsMat.wrap_torch((n, n), b_ip, b_ix, b_dt)
    stream = torch.cuda.current_stream()
    c = smmp.mul_csr_csr(a, b)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for p_, q_ in ev:
        c = None
        p_.record(stream)
        c = smmp.mul_csr_csr(a, b)
        q_.record(stream)
    torch.cuda.synchronize()
    ms = [p_.elapsed_time(q_) for p_, q_ in ev]
    sec = float(np.mean(ms)) * 1e-3
    nnz_a, nnz_b, nnz_c = int(a_ix.numel()), int(b_ix.numel()), int(c.nnz())
    b_len = (b_ip[1:] - b_ip[:-1]).to(torch.float64)
    products = float(b_len[a_ix.long()].sum().item())
    comp = (nnz_a + nnz_b + nnz_c) * 16 + 3 * (n + 1) * 8
    out = {"workload": "sprs-benches shape bench, largest shape: (%d x %d) * (%d x %d), uniform density %d / cols, values N(0,1), usize indices"
                       % (n, n, n, n, nnz_over_rows),
           "seconds_per_product": round(sec, 6), "kernel_ms_min": round(float(np.min(ms)), 3), "steps": steps,
           "nnz_a": nnz_a, "nnz_b": nnz_b, "nnz_c": nnz_c, "products": products, "gflops": round(2 * products / sec / 1e9, 2),
           "compulsory_bytes": comp, "achieved_GBs": round(comp / sec / 1e9, 1), "roofline_frac": round(comp / sec / 1e9 / HBM_PEAK_GBS, 4)}
    if with_cpu:
        from oracle import oracle   # test infrastructure: the checker + the timed CPU port, never the product
        h = lambda t: t.cpu().numpy().view(np.uint64)
        args = ((n, n), h(a_ip), h(a_ix), a_dt.cpu().numpy(), (n, n), h(b_ip), h(b_ix), b_dt.cpu().numpy())
        t = time.perf_counter()
        _, r_ip, r_ix, r_dt = oracle.mul_csr_csr(*args, threads=1)
        t1 = time.perf_counter() - t
        t = time.perf_counter()
        res = oracle.mul_csr_csr(*args, threads=0, return_threads=True)
        tauto = time.perf_counter() - t
        _, g_ip, g_ix, g_dt = c.to_host()
        same = bool(np.array_equal(g_ip, r_ip) and np.array_equal(g_ix, r_ix))
        bits = bool(same and np.array_equal(g_dt.view(np.uint64), r_dt.view(np.uint64)))
        same_auto = bool(np.array_equal(res[1], r_ip) and np.array_equal(res[2], r_ix) and np.array_equal(res[3].view(np.uint64), r_dt.view(np.uint64)))
        out["parity"] = {"entries_checked": nnz_c, "structure_bit_exact": same, "values_bit_exact": bits,
                         "oracle_auto_equals_fixed1": same_auto,          # the reference's own assert_eq!(prod, prod_) (main.rs:228, 242, 256)
                         "tolerance": 1e-10, "ok": bool(same and bits)}
        out["cpu_baseline"] = {"fixed1_seconds": round(t1, 3), "automatic_seconds": round(tauto, 3), "automatic_threads": int(res[-1]),
                               "kind": "port", "host_cores": oracle.num_procs(),
                               "sample": "the whole product, once per strategy (C restatement of smmp::mul_csr_csr; rustc is not available here)"}
    return out


def sparse_dense_micro(dev, with_oracle=True):
    """The reference's own micro-benchmark of the dense product (sprs/benches/sparse_dense_products.rs:21-55): a 3 x 1 000 000
    CsMat with 5 stored entries times the dense vector Array::range(0., 10., 0.00001) — `&a * &w` and the specialised
    csr_mulacc_dense_colmaj call are the same kernel here.  Five multiply-adds: on a GPU this is one launch's latency, reported
    as such beside the oracle's time for the same call (it is not a shape to offload; the line says so instead of hiding it)."""
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    n = 1_000_000
    w = torch.arange(n, dtype=torch.float64, device=dev) * 0.00001
    ip = np.array([0, 2, 4, 5], dtype=np.uint64)
    ix = np.array([0, 1, 0, 2, 2], dtype=np.uint64)
    dt = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    a = DeviceCsMat.from_host((3, n), ip, ix, dt)
    y = torch.empty(3, dtype=torch.float64, device=dev)
    wv, yv = DeviceVec.borrow(w), DeviceVec.borrow(y)
    stream = torch.cuda.current_stream()
    for _ in range(5):
        prod.csmat_mul_vec(a, wv, out=yv, stream=stream)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    t = time.perf_counter()
    for p_, q_ in ev:
        p_.record(stream)
        prod.csmat_mul_vec(a, wv, out=yv, stream=stream)
        q_.record(stream)
    torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t) / len(ev) * 1e6
    us = [p_.elapsed_time(q_) * 1e3 for p_, q_ in ev]
    out = {"workload": "sprs/benches/sparse_dense_products.rs: (3 x 1e6 CsMat, 5 entries) * dense vec of 1e6", "gpu_us_per_call_events": round(float(np.mean(us)), 2),
           "gpu_us_per_call_wall": round(wall_us, 2), "note": "launch-latency bound: five multiply-adds; not a shape to offload"}
    if with_oracle:
        from oracle import oracle   # test infrastructure: the checker + the timed CPU port, never the product
        w_h = w.cpu().numpy()
        y_h = np.zeros(3)
        reps = 2000
        t = time.perf_counter()
        for _ in range(reps):
            y_h[:] = 0.0
            oracle.mul_acc_mat_vec_csr((3, n), ip, ix, dt, w_h, y_h)
        out["cpu_us_per_call"] = round((time.perf_counter() - t) / reps * 1e6, 2)       # includes the ctypes call overhead of the harness
        out["parity"] = {"bit_exact": bool(np.array_equal(y.cpu().numpy(), y_h)), "ok": bool(np.array_equal(y.cpu().numpy(), y_h))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="rmat10m")
    ap.add_argument("--idx-bytes", type=int, default=8, choices=(4, 8))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="lib", choices=("torch", "lib", "peer"),
                    help="N > 1: all-gather-v of y inside the library over RCCL (sprs_hip_dist_*: sub-block pipeline, grouped ncclSend / "
                         "ncclRecv; default, what `value` times), inside the library by stores into the peers' windows over xGMI (peer: "
                         "hand-written, no RCCL call on the data path) or through torch.distributed (grouped send/recv on RCCL); the "
                         "other routes and the multiply without any exchange are timed after the K steps and reported beside it")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="N > 1: torch.distributed backend.  nccl = RCCL over xGMI (the product path).  gloo: the y blocks are staged "
                         "through host memory — for running the N > 1 code path (self-launch, partition, RowShardedSpMV on the HIP "
                         "kernels, route agreement, JSON keys) where RCCL cannot come up, e.g. N ranks on ONE GPU with --devices 0,0; the "
                         "library's own RCCL route is not attempted then and the line says so")
    ap.add_argument("--devices", default="",
                    help="N > 1: comma-separated device of every local rank (default: rank i on device i); ranks may share a device with --backend gloo")
    ap.add_argument("--no-secondary", action="store_true", help="default workload only: skip the short spgemm5 object")
    ap.add_argument("--permute-cols", type=int, default=0,
                    help="experiment: relabel the columns by a random permutation (seed given) before the run")
    ap.add_argument("--kernel", type=int, default=None, help="spmv_kernel option for A/B (1 tiled, 2 wave-per-row)")
    ap.add_argument("--xcs", type=int, default=None, help="XCD-sliced plan: 0 auto, 1 on, 2 off")
    ap.add_argument("--split", type=int, default=None, help="row-length threshold of the sliced part")
    ap.add_argument("--idx32", type=int, default=None, help="plan copies with 32-bit column ids: 1 on (default), 0 off")
    ap.add_argument("--sort", type=int, default=None, help="plan copies with column-sorted tiles: 1 on (default), 0 off")
    ap.add_argument("--tile", type=int, default=None, help="nnz per workgroup tile: 2048 or 4096")
    ap.add_argument("--relabel", type=int, default=None, help="sliced plan column relabelling: 0 auto, 1 on, 2 off")
    ap.add_argument("--ldspad", type=int, default=None, help="extra dynamic LDS per workgroup (occupancy cap, tuning)")
    ap.add_argument("--band", type=int, default=None, help="banded plan (hot columns from LDS): 0 auto, 1 on, 2 off")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="any library option (sprs_hip_set_option) for an A/B, e.g. --opt spgemm_micro=3; repeatable")
    ap.add_argument("--cold-cache", action="store_true",
                    help="stream a 1 GiB scratch buffer between steps (outside the per-step kernel events): the matrix and x "
                         "then come from HBM, not from the 256 MiB Infinity Cache (SURVEY 8d, config 2)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    dev_index = local_rank
    if args.devices:
        devs = [int(v) for v in args.devices.split(",")]
        if len(devs) != world:
            sys.exit("--devices names %d devices for %d ranks" % (len(devs), world))
        if len(set(devs)) != len(devs) and args.backend == "nccl":
            sys.exit("ranks sharing a device need --backend gloo (RCCL refuses duplicate devices)")
        dev_index = devs[local_rank]
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # small control tensors of the collectives: on the device for RCCL, on the host for gloo
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import sprs_amd
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.dist import RowShardedSpMV
    from sprs_amd import _ffi
    import ctypes as C
    _ffi.check(_ffi.lib.sprs_hip_set_device(dev_index))
    for opt, val in (("spmv_kernel", args.kernel), ("spmv_xcs", args.xcs), ("spmv_xcs_split", args.split), ("spmv_xcs_idx32", args.idx32), ("spmv_sort_tiles", args.sort), ("spmv_tile", args.tile), ("spmv_relabel", args.relabel), ("spmv_lds_pad", args.ldspad), ("spmv_band", args.band)):
        if val is not None:
            sprs_amd.set_option(opt, val)
    for nv in args.opt:
        sprs_amd.set_option(nv.split("=")[0], int(nv.split("=")[1]))

    if args.workload == "spgemm_uniform":
        if world != 1:
            sys.exit("spgemm_uniform is single-GPU (north star: SpGEMM stays on one GPU)")
        out = spgemm_uniform(dev, steps=args.steps if args.steps != 50 else 5, with_cpu=not args.no_cpu_baseline)
        out.update({"metric": "CSR x CSR SpGEMM seconds per product (sprs-benches shape)", "value": out["seconds_per_product"], "unit": "s",
                    "higher_is_better": False, "n_gpus": 1, "scaling": "replicas only", "vs_baseline": None, "dtype": "f64", "data": "synthetic"})
        print(json.dumps(out))
        return
    if args.workload == "spgemm5":
        if world != 1:
            sys.exit("spgemm5 is single-GPU (north star: SpGEMM stays on one GPU)")
        steps = args.steps if args.steps != 50 else 5
        out = spgemm5(dev, args.idx_bytes, steps, min(args.warmup, 2), check_rows=100, cpu_blocks=16, cpu_block_rows=1500,
                      with_cpu=not args.no_cpu_baseline)
        out.update({"n_gpus": 1, "scaling": "replicas only", "vs_baseline": None})
        print(json.dumps(out))
        return

    # ---- workload -----------------------------------------------------------
    idt = torch.int64 if args.idx_bytes == 8 else torch.int32
    t0 = time.time()
    wl = args.workload
    if wl == "rmat10m":
        n, k = 10_000_000, 32
        indptr, indices, data = gen.rmat_csr(n, k, device=dev, idx_dtype=idt, ptr_dtype=idt)
        name = "R-MAT 10M x 10M, ~32 nnz/row (Graph500 a,b,c,d=.57,.19,.19,.05; seed 1)"
    elif wl == "rmat1m":
        n, k = 1_000_000, 16
        indptr, indices, data = gen.rmat_csr(n, k, device=dev, idx_dtype=idt, ptr_dtype=idt)
        name = "R-MAT 1M x 1M, ~16 nnz/row"
    elif wl == "laplace4096":
        n = 4096 * 4096
        indptr, indices, data = gen.grid_laplacian(4096, 4096, device=dev, idx_dtype=idt, ptr_dtype=idt)
        name = "5-pt Laplacian 4096^2 grid"
    elif wl.startswith("rmat:"):
        _, n, k = wl.split(":")
        n, k = int(n), float(k)
        indptr, indices, data = gen.rmat_csr(n, k, device=dev, idx_dtype=idt, ptr_dtype=idt)
        name = "R-MAT %d x %d, ~%g nnz/row" % (n, n, k)
    else:
        sys.exit("unknown workload " + wl)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    if args.permute_cols:
        # same sparsity statistics, but the hub columns no longer sit at 0, 2^k, 2^j + 2^k: separates what the
        # power-law costs from what the ADDRESSES of its hot x entries cost (L2 channel hot-spotting)
        if args.permute_cols > 0:
            g = torch.Generator(device=dev)
            g.manual_seed(args.permute_cols)
            perm = torch.randperm(n, device=dev, generator=g).to(indices.dtype)
        elif args.permute_cols <= -2:   # -K: the K hottest columns move to the front (natural order inside), rest follows
            cnt = torch.bincount(indices.long(), minlength=n)
            thr = torch.sort(cnt, descending=True).values[-args.permute_cols]
            hot = cnt > thr
            order_c = torch.cat([torch.nonzero(hot).flatten(), torch.nonzero(~hot).flatten()])
            perm = torch.empty(n, dtype=indices.dtype, device=dev)
            perm[order_c] = torch.arange(n, device=dev, dtype=indices.dtype)
            del cnt, order_c, hot
        else:   # -1: relabel by decreasing column count (hub columns become 0, 1, 2, ...)
            cnt = torch.bincount(indices.long(), minlength=n)
            order_c = torch.argsort(cnt, descending=True, stable=True)
            perm = torch.empty(n, dtype=indices.dtype, device=dev)
            perm[order_c] = torch.arange(n, device=dev, dtype=indices.dtype)
            del cnt, order_c
        indices = perm[indices.long()]
        # rows must stay sorted by column (CsMat invariant): sort inside rows via a (row, col) key
        rows_of = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
        key = (rows_of << 32) | indices.long()
        if os.environ.get("BENCH_NO_RESORT"):   # experiment: keep the entries in their old order (rows no longer sorted)
            order = torch.arange(key.numel(), device=dev)
        else:
            key, order = torch.sort(key)
        indices = (key & 0xFFFFFFFF).to(idt)
        data = data[order]
        del rows_of, key, order, perm
        name += " [columns permuted]" if args.permute_cols > 0 else " [columns relabelled by decreasing count]"
    nnz_total = indices.numel()
    x = gen.dense_vector(n, seed=3, device=dev)
    stream = torch.cuda.current_stream()

    handles = {}
    prep_s = [0.0]

    def local_spmv(block, xv, y_block):
        key = id(block)
        if key not in handles:
            rows_b, cols_b, ip, ix, dt = block
            handles[key] = DeviceCsMat.wrap_torch((rows_b, cols_b), ip, ix, dt)
            if world > 1:
                # N > 1: every route's handle gets its final plan up front (plan policy, sprs_hip.h) — the routes cross-check and the
                # secondary timings call a handle once or twice outside any warm-up, and a deferred plan build must not land in them
                torch.cuda.synchronize()
                t_prep = time.perf_counter()
                handles[key].prepare(stream=stream)
                torch.cuda.synchronize()
                prep_s[0] += time.perf_counter() - t_prep
        out = DeviceVec.borrow(y_block)
        prod.csmat_mul_vec(handles[key], DeviceVec.borrow(xv), out=out, stream=stream)

    sh = RowShardedSpMV((n, n), indptr, indices, data, local_spmv)
    libdist = None

    def torch_step(xv):                                 # multiply, then torch.distributed's grouped send/recv
        sh.local_spmv(sh.block, xv, sh.y[sh.r0:sh.r1])
        sh.exchange()
        return sh.y

    def multiply_only(xv):
        sh.local_spmv(sh.block, xv, sh.y[sh.r0:sh.r1])
        return sh.y

    step = torch_step
    peerdist = None                                     # the library handle on the peer-store route (its own handle: the route is per handle)
    if world > 1:
        from sprs_amd.dist import DistSpMV
        rb = sh.block
        yv_all = DeviceVec.borrow(sh.y)
        if args.backend != "nccl":
            if args.exchange == "lib" and rank == 0:
                print("bench.py: --backend gloo: the library's RCCL exchange is not attempted; timing the torch.distributed route "
                      "(y blocks staged through host memory)", file=sys.stderr)
        else:
            try:
                libdist = DistSpMV((n, n), DeviceCsMat.wrap_torch((rb[0], rb[1]), rb[2], rb[3], rb[4]), sh.cuts, rank, world,
                                   unique_id=DistSpMV.broadcast_id(dev), nsub=2)
            except Exception as e:                      # the library route needs librccl: say so, fall back to the torch route
                if args.exchange == "lib" and rank == 0:
                    print("bench.py: library exchange unavailable (%s); timing the torch.distributed route" % repr(e)[:200], file=sys.stderr)
                libdist = None
        # the peer-store route needs no RCCL: window handles through torch.distributed (any backend), also for ranks sharing a device.
        # Every step that can fail on ONE rank is followed by an agreement of all ranks, so that nobody waits in a collective alone.
        import torch.distributed as dist
        perr = None
        try:
            peerdist = DistSpMV((n, n), DeviceCsMat.wrap_torch((rb[0], rb[1]), rb[2], rb[3], rb[4]), sh.cuts, rank, world,
                                unique_id=None, nsub=2)
        except Exception as e:
            perr, peerdist = e, None
        made = torch.tensor([1.0 if peerdist is not None else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(made, op=dist.ReduceOp.MIN)
        if float(made.item()) > 0.5:
            try:
                peerdist.connect_peers(dev).set_route("peer")      # (collective-safe itself: sprs_amd/dist.py)
            except Exception as e:
                perr, peerdist = e, None
        else:
            peerdist = None
        if peerdist is None and rank == 0:
            print("bench.py: peer-store exchange unavailable (%s)" % repr(perr)[:200], file=sys.stderr)

    def lib_step(xv):                                   # multiply + exchange inside the library, sub-blocks pipelined
        libdist.spmv(DeviceVec.borrow(xv), yv_all, stream=stream)
        return sh.y

    def peer_step(xv):                                  # the same with stores into the peers' windows instead of ncclSend / ncclRecv
        peerdist.spmv(DeviceVec.borrow(xv), yv_all, stream=stream)
        return sh.y

    routes_check = None
    if world > 1:
        # every rank must take the same route: the library route only if it came up on ALL ranks, and only if one step through it
        # gives what one step through torch.distributed gives (the two exchanges move the same blocks; the multiply is the same
        # kernel) — otherwise all ranks time the torch route together and the line says so
        import torch.distributed as dist
        flag = torch.tensor([1.0 if libdist is not None else 0.0, 1.0 if peerdist is not None else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag[0].item()) < 0.5:
            libdist = None
        if float(flag[1].item()) < 0.5:
            peerdist = None
        routes_check = {}
        for label, have, fn in (("lib", libdist is not None, lib_step), ("peer", peerdist is not None, peer_step)):
            if not have:
                continue
            worst = torch.tensor([float("inf")], dtype=torch.float64, device=dev)
            try:
                y_t = torch_step(x).clone()
                sh.y.fill_(float("nan"))                # the route under test must write every row itself
                y_l = fn(x)
                torch.cuda.synchronize()
                worst = ((y_l - y_t).abs() / y_t.abs().clamp_min(1e-300)).max().reshape(1)
                worst = torch.where(torch.isfinite(worst), worst, torch.full_like(worst, float("inf")))
                del y_t
            except Exception as e:
                if rank == 0:
                    print("bench.py: %s exchange failed in the cross-check (%s)" % (label, repr(e)[:200]), file=sys.stderr)
            worst = worst.to(cdev)
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            routes_check[label + "_vs_torch_max_rel_diff"] = float(worst.item())
            if not float(worst.item()) <= 1e-10:
                if label == "lib":
                    libdist = None
                else:
                    peerdist = None
        if routes_check:
            routes_check.update({"tolerance": 1e-10, "ok": bool(all(v <= 1e-10 for k, v in routes_check.items() if k.endswith("_diff")))})
        else:
            routes_check = None
    use_lib = (libdist is not None and args.exchange == "lib") or (peerdist is not None and args.exchange == "peer")
    if use_lib:
        step = lib_step if args.exchange == "lib" else peer_step
        lib_step_timed = step
    del indptr, indices, data   # only the rank's block stays resident
    torch.cuda.empty_cache()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # plan policy (include/sprs_hip.h, sprs_hip_csmat_prepare): the FIRST multiply of a handle runs on the plain tile index over its own
    # arrays, the re-laid-out copy (banded plan) is built by the SECOND — both are timed here, outside the timed region
    torch.cuda.synchronize()
    t_plan = time.perf_counter()
    step(x)
    torch.cuda.synchronize()
    first_step_s = time.perf_counter() - t_plan
    first_plan = handles[id(sh.block)].spmv_plan_info() if handles else (0, 0)
    t_plan = time.perf_counter()
    step(x)                                             # the second multiply builds the copy plan cached in the handle
    torch.cuda.synchronize()
    second_step_s = time.perf_counter() - t_plan
    for _ in range(args.warmup):
        step(x)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps ----------------------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    flush = torch.empty(1 << 27, dtype=torch.float64, device=dev) if args.cold_cache else None   # 1 GiB
    t_start = time.perf_counter()
    for s in range(args.steps):
        if flush is not None:
            flush.add_(1.0)   # reads and writes 1 GiB: evicts L2 and the Infinity Cache; part of ms_per_step, not of the kernel events
        ev[s][0].record(stream)
        if use_lib:
            lib_step_timed(x)                           # multiply + exchange inside the library (RCCL or peer stores), pipelined
            ev[s][1].record(stream)
        else:
            sh.local_spmv(sh.block, x, sh.y[sh.r0:sh.r1])   # kernel(s) on `stream`, bracketed by HIP events
            ev[s][1].record(stream)
            sh.exchange()
    torch.cuda.synchronize()
    barrier()
    t_total = time.perf_counter() - t_start

    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([t_total], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_total = float(tt.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_ms = float(np.mean(kern_ms))

    # ---- N > 1: the other exchange route and the multiply alone, after the K timed steps (with and without the gather) ----
    exchange_times = None
    if world > 1:
        import torch.distributed as dist

        def timed_ms(fn, k):
            fn(x)                                       # one untimed call: no first-use cost of a route inside its timing
            torch.cuda.synchronize()
            barrier()
            t = time.perf_counter()
            for _ in range(k):
                fn(x)
            torch.cuda.synchronize()
            barrier()
            tt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return round(float(tt.item()) * 1e3 / k, 5)
        k2 = max(3, args.steps // 2)
        exchange_times = {"timed_route": args.exchange if use_lib else "torch", "steps_each": k2, "backend": args.backend,
                          "devices": args.devices or ",".join(str(i) for i in range(world))}
        if libdist is not None:
            try:
                exchange_times["rccl_comm_ranks"] = libdist.comm_count()     # what the communicator itself says (ncclCommCount)
            except Exception as e:
                exchange_times["rccl_comm_ranks"] = "failed: " + repr(e)[:120]
        for label, fn in (("multiply_only_ms", multiply_only), ("torch_route_ms", torch_step),
                          ("lib_route_ms", lib_step if libdist is not None else None),
                          ("peer_route_ms", peer_step if peerdist is not None else None)):
            try:
                exchange_times[label] = timed_ms(fn, k2) if fn is not None else None
            except Exception as e:                      # the headline line must not depend on a secondary measurement
                exchange_times[label] = "failed: " + repr(e)[:120]
        if isinstance(exchange_times.get("multiply_only_ms"), float) and use_lib:
            kern_avg_ms = exchange_times["multiply_only_ms"]      # the roofline object prices the local multiply, not the exchange

    ms_per_step = t_total * 1e3 / args.steps
    gflops = 2.0 * nnz_total * args.steps / t_total / 1e9
    blk_rows, blk_cols = sh.block[0], sh.block[1]
    alg_bytes = algorithmic_bytes(blk_rows, blk_cols, sh.block_nnz, args.idx_bytes, args.idx_bytes)
    achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9

    # PMC counters cannot be collected inside this process: `traffic` comes from the committed
    # counter passes of the same command (scripts/gpu_pmc.sh -> profiles/pmc_traffic.json) and is
    # only filled in when workload, index width and options match that run.
    traffic = None
    defaults = all(v is None for v in (args.kernel, args.xcs, args.split, args.idx32, args.sort, args.tile, args.ldspad, args.relabel, args.band)) and not args.permute_cols
    try:
        if world == 1 and defaults and not args.cold_cache:
            sha = csrc_sha16("spmv")
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                for e in json.load(f)["entries"]:
                    # only a measurement taken on exactly these kernel sources counts (stale numbers read as null)
                    if e["workload"] == wl and e["index_bytes"] == args.idx_bytes and e.get("csrc_sha16") == sha:
                        traffic = e["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        traffic = None
    plan_kind, plan_bytes = handles[id(sh.block)].spmv_plan_info() if handles else (0, 0)
    kernel_name = {3: "all kernels of one SpMV on the banded plan: band_gather (x into the plan's labelling) + band_hot (x tile in LDS) + "
                      "band_cold (cold pieces, short rows) + band_reduce (small plans: band_tail = reduction + short rows)",
                   2: "all kernels of one SpMV on the XCD-sliced plan: rl_permute_x + spmv_tile (short rows) + spmv_sliced + carry / reduce kernels",
                   1: "sprs_hip::spmv_tile_kernel (+ spmv_carry_kernel)"}.get(plan_kind, "sprs_hip::spmv_rowwave_kernel")
    out = {
        "metric": "CSR SpMV GFLOP/s",
        "value": round(gflops, 3),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": name,
            "rows": n, "cols": n, "nnz": nnz_total,
            "index_bytes": args.idx_bytes, "indptr_bytes": args.idx_bytes,
            "partition": ("cost-balanced (nnz + %g/row) contiguous row blocks x%d, direct all-gather-v of y (%s)" %
                          (sh.row_weight, world, ("sprs_hip_dist_*, RCCL inside the library" if args.exchange == "lib" else
                                                  "sprs_hip_dist_*, stores into the peers' windows (no RCCL on the data path)") if use_lib else
                           "torch.distributed grouped send/recv on RCCL" if args.backend == "nccl" else "torch.distributed send/recv on gloo, staged through host memory"))
                         if world > 1 else "single GPU",
            "generate_s": round(gen_s, 2),
            # once per handle, never part of `value`: the first multiply (plan build + one SpMV) minus a steady-state step
            # (N > 1: every handle is prepared when it is made — the time of that sprs_hip_csmat_prepare call)
            "plan_build_s": round(max(0.0, second_step_s - ms_per_step * 1e-3), 4) if world == 1 else round(prep_s[0], 4),
            # the handle's FIRST multiply (plain tile index over the handle's own arrays + one SpMV on it): what a caller that
            # multiplies once pays instead of plan_build_s
            "first_spmv_ms": round(first_step_s * 1e3, 3),
            "first_spmv_plan": {0: "none", 1: "nnz tiles", 2: "xcd-sliced copy", 3: "banded copy"}.get(first_plan[0], "?"),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms_avg": round(kern_avg_ms, 5),
            "kernel_ms_min": round(float(np.min(kern_ms)), 5),
            "traffic": traffic,   # HBM bytes per SpMV from the committed rocprofv3 --pmc passes of THESE kernel sources
                                  # (profiles/pmc_traffic.json, matched by csrc_sha16), else null
            "plan": {1: "nnz tiles", 2: "xcd-sliced copy", 3: "banded copy (hot columns from LDS)"}.get(plan_kind, "none"),
            "plan_bytes": plan_bytes,
            "cold_cache": bool(args.cold_cache),
        },
    }

    if exchange_times is not None:
        if routes_check is not None:
            exchange_times["routes_agree"] = routes_check
        out["exchange"] = exchange_times

    # ---- N > 1: parity of the distributed result (every rank checks ITS row block against the oracle on that block, and that the
    # gathered y is the same vector on every rank) — the checker only, after the timed region ------------------------------------
    if world > 1 and not args.no_cpu_baseline:
        import torch.distributed as dist
        try:
            from oracle import oracle   # test infrastructure: the checker, never the product
            npdt = np.uint64 if args.idx_bytes == 8 else np.uint32
            step(x)
            torch.cuda.synchronize()
            y_all = sh.y.cpu().numpy()
            yb = np.zeros(sh.block[0])
            oracle.mul_acc_mat_vec_csr((sh.block[0], n), sh.block[2].cpu().numpy().view(npdt), sh.block[3].cpu().numpy().view(npdt),
                                       sh.block[4].cpu().numpy(), x.cpu().numpy(), yb)
            got = y_all[sh.r0:sh.r1]
            den = np.maximum(np.abs(yb), np.abs(got))
            rel = float(np.max(np.where(den > 0, np.abs(got - yb) / np.where(den > 0, den, 1.0), 0.0))) if den.size else 0.0
            chk = float(np.sum(y_all * (1.0 + (np.arange(n) % 7))))        # position-weighted checksum of the gathered vector
        except Exception as e:
            if rank == 0:
                print("bench.py: parity of the distributed result failed to run (%s)" % repr(e)[:200], file=sys.stderr)
            rel, chk = float("inf"), float(rank)
        t_rel = torch.tensor([rel], dtype=torch.float64, device=cdev)
        t_lo, t_hi = torch.tensor([chk], dtype=torch.float64, device=cdev), torch.tensor([chk], dtype=torch.float64, device=cdev)
        dist.all_reduce(t_rel, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(t_hi, op=dist.ReduceOp.MAX)
        same = bool(float(t_lo.item()) == float(t_hi.item()))
        out["parity"] = {"max_rel_err_vs_oracle": float(t_rel.item()), "tolerance": 1e-10, "what": "every rank: its row block of y against the oracle on "
                         "that block (max over ranks); gathered_y_identical: the position-weighted checksum of the whole y is the same double on every rank",
                         "gathered_y_identical": same, "ok": bool(float(t_rel.item()) <= 1e-10 and same)}

    # ---- CPU baseline (rank 0, N = 1): the oracle on the host cores -------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle   # test infrastructure: the checker + the timed CPU port, never the product
        npdt = np.uint64 if args.idx_bytes == 8 else np.uint32
        ip_h = sh.block[2].cpu().numpy().view(npdt)
        ix_h = sh.block[3].cpu().numpy().view(npdt)
        dt_h = sh.block[4].cpu().numpy()
        x_h = x.cpu().numpy()
        y_gpu = sh.y.cpu().numpy()
        reps = 3
        ts = []
        for _ in range(reps):
            y_h = np.zeros(n)
            t = time.perf_counter()
            oracle.mul_acc_mat_vec_csr((n, n), ip_h, ix_h, dt_h, x_h, y_h)
            ts.append(time.perf_counter() - t)
        serial_s = min(ts)
        denom = np.maximum(np.abs(y_h), np.abs(y_gpu))
        rel = np.where(denom > 0, np.abs(y_gpu - y_h) / np.where(denom > 0, denom, 1.0), 0.0)
        out["cpu_baseline"] = {
            "value": round(2.0 * nnz_total / serial_s / 1e9, 4),
            "unit": "GFLOP/s",
            "cores": 1,
            "kind": "port",
            "sample": "whole %s matrix, best of %d SpMVs; C restatement of sprs' serial prod::mul_acc_mat_vec_csr "
                      "(sprs SpMV is single-threaded; rustc is not available here)" % (wl, reps),
            "seconds": round(serial_s, 4),
        }
        out["parity"] = {"max_rel_err_vs_oracle": float(rel.max()), "tolerance": 1e-10,
                         "ok": bool(rel.max() <= 1e-10)}
        # NOT in the reference (sprs' SpMV is serial): the same loop split over the host's cores by rows with OpenMP, for scale
        try:
            ncpu = os.cpu_count() or 1
            ts = []
            for _ in range(2):
                y_o = np.zeros(n)
                t = time.perf_counter()
                oracle.mul_acc_mat_vec_csr((n, n), ip_h, ix_h, dt_h, x_h, y_o, threads=ncpu)
                ts.append(time.perf_counter() - t)
            out["cpu_baseline"]["all_cores_openmp"] = {"value": round(2.0 * nnz_total / min(ts) / 1e9, 3), "unit": "GFLOP/s", "cores": ncpu,
                                                        "seconds": round(min(ts), 4), "note": "not in reference: OpenMP row split of the same loop"}
        except Exception as e:
            out["cpu_baseline"]["all_cores_openmp"] = {"error": repr(e)[:160]}
        # PCIe-inclusive note (never part of `value`): the host arrays uploaded once through the boundary's host entry
        try:
            torch.cuda.synchronize()
            t = time.perf_counter()
            up = DeviceCsMat.from_host((n, n), ip_h, ix_h, dt_h)
            _ffi.check(_ffi.lib.sprs_hip_synchronize(None))
            out["config"]["h2d_upload_s"] = round(time.perf_counter() - t, 4)
            del up
        except Exception as e:
            out["config"]["h2d_upload_s"] = "failed: " + repr(e)[:120]

    # ---- the same matrix with u32 indices and indptr (a second line, 12 B per entry, never mixed with the headline) ----
    if rank == 0 and world == 1 and wl == "rmat10m" and args.idx_bytes == 8 and not args.no_secondary:
        try:
            ip4, ix4, dt4 = gen.rmat_csr(n, 32, device=dev, idx_dtype=torch.int32, ptr_dtype=torch.int32)
            a4 = DeviceCsMat.wrap_torch((n, n), ip4, ix4, dt4)
            y4 = DeviceVec.borrow(torch.empty(n, dtype=torch.float64, device=dev))
            xv = DeviceVec.borrow(x)
            for _ in range(3):
                prod.csmat_mul_vec(a4, xv, out=y4, stream=stream)
            torch.cuda.synchronize()
            k4 = max(5, args.steps // 2)
            ev4 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k4)]
            for p_, q_ in ev4:
                p_.record(stream)
                prod.csmat_mul_vec(a4, xv, out=y4, stream=stream)
                q_.record(stream)
            torch.cuda.synchronize()
            ms4 = float(np.mean([p_.elapsed_time(q_) for p_, q_ in ev4]))
            nnz4 = int(ix4.numel())
            b4 = algorithmic_bytes(n, n, nnz4, 4, 4)
            pk4, pb4 = a4.spmv_plan_info()
            out["spmv_u32"] = {"workload": name + ", u32 indices and indptr", "nnz": nnz4, "steps": k4, "kernel_ms_avg": round(ms4, 5),
                               "gflops": round(2.0 * nnz4 / (ms4 * 1e-3) / 1e9, 2), "algorithmic_bytes_per_launch": b4,
                               "achieved_GBs": round(b4 / (ms4 * 1e-3) / 1e9, 2), "frac": round(b4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "plan_bytes": pb4,
                               "note": "the banded plan stores its own 16-bit / 32-bit ids whatever the handle's index width, so a u32 handle streams "
                                       "the SAME %.2f GB plan as the usize handle and takes the same time; `frac` here prices that time against the "
                                       "smaller 12 B/entry CSR formula, which is why it reads lower than the headline — the narrower indices save "
                                       "device memory for the handle (1.3 GB) and upload time, not SpMV time" % (pb4 / 1e9)}
            del a4, ip4, ix4, dt4, y4
            torch.cuda.empty_cache()
        except Exception as e:   # the headline line must not depend on the secondary measurement
            out["spmv_u32"] = {"error": repr(e)[:200]}

    # ---- SpMM on the headline matrix (prod::csr_mulacc_dense_rowmaj, k = 8 and 16 rhs columns) -----------------------
    if rank == 0 and world == 1 and wl == "rmat10m" and args.idx_bytes == 8 and not args.no_secondary and handles:
        try:
            import ctypes as C
            a8 = handles[id(sh.block)]
            out["spmm"] = {"workload": name + ", rhs and result dense row-major (&CsMat * &Array2 with >= 8 columns)", "k": {}}
            for kk in (8, 16):
                rhs = gen.dense_vector(n * kk, seed=5, device=dev)
                res = torch.empty(n * kk, dtype=torch.float64, device=dev)
                call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(a8._h, C.c_void_p(rhs.data_ptr()), n, kk, kk, C.c_void_p(res.data_ptr()),
                                                                            n, kk, 0, C.c_void_p(stream.cuda_stream)))
                for _ in range(2):
                    call()
                torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for p_, q_ in evs:
                    p_.record(stream)
                    call()
                    q_.record(stream)
                torch.cuda.synchronize()
                msk = float(np.mean([p_.elapsed_time(q_) for p_, q_ in evs]))
                # algorithmic bytes: the CSR arrays once, the rhs once, the result once (scripts/spmm_bench.py)
                algk = nnz_total * 16 + (n + 1) * 8 + 2 * n * kk * 8
                # parity of column 0 against the SpMV (itself checked against the oracle above)
                x0 = rhs.view(n, kk)[:, 0].contiguous()
                y0 = torch.empty(n, dtype=torch.float64, device=dev)
                prod.csmat_mul_vec(a8, DeviceVec.borrow(x0), out=DeviceVec.borrow(y0), stream=stream)
                torch.cuda.synchronize()
                err = float(((res.view(n, kk)[:, 0] - y0).abs() / y0.abs().clamp_min(1e-300)).max())
                # ... and a row sample of TWO columns against the oracle itself (the checker; SpMM == SpMV per rhs column, prod.rs:274-298)
                orc = None
                if not args.no_cpu_baseline and "ip_h" in locals():
                    from oracle import oracle   # test infrastructure: the checker, never the product
                    worst, checked = 0.0, 0
                    for r_lo in (0, n // 2, n - 60000):
                        r_hi = r_lo + 60000
                        lo_, hi_ = int(ip_h[r_lo]), int(ip_h[r_hi])
                        ipb = (ip_h[r_lo:r_hi + 1] - ip_h[r_lo]).astype(ip_h.dtype)
                        for col in (0, kk - 1):
                            xc = rhs.view(n, kk)[:, col].contiguous().cpu().numpy()
                            yb = np.zeros(r_hi - r_lo)
                            oracle.mul_acc_mat_vec_csr((r_hi - r_lo, n), ipb, ix_h[lo_:hi_], dt_h[lo_:hi_], xc, yb)
                            got = res.view(n, kk)[r_lo:r_hi, col].cpu().numpy()
                            den = np.maximum(np.abs(yb), np.abs(got))
                            worst = max(worst, float(np.max(np.where(den > 0, np.abs(got - yb) / np.where(den > 0, den, 1.0), 0.0))))
                        checked += r_hi - r_lo
                    orc = {"rows_checked": checked, "columns_checked": [0, kk - 1], "max_rel_err": worst, "tolerance": 1e-10, "ok": bool(worst <= 1e-10)}
                out["spmm"]["k"][str(kk)] = {"vs_oracle": orc,"kernel_ms_avg": round(msk, 4), "gflops": round(2.0 * nnz_total * kk / (msk * 1e-3) / 1e9, 1),
                                             "algorithmic_bytes_per_launch": algk, "achieved_GBs": round(algk / (msk * 1e-3) / 1e9, 1),
                                             "frac": round(algk / (msk * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                             "rhs_rows_gathered_per_s": round(nnz_total / (msk * 1e-3) / 1e9, 1),
                                             "col0_vs_spmv_max_rel": err, "tolerance": 1e-10, "ok": bool(err <= 1e-10)}
                del rhs, res
            out["spmm"]["note"] = ("bound: one 128-byte fabric request per stored entry (the rhs row of its column); random 128-byte gathers top out at "
                                   "57.5 G/s on this part (scripts/probes/hbm_patterns.hip, profiles/r11g_hbm_patterns.jsonl); the rhs is gathered from a "
                                   "re-laid-out copy (option spmm_relayout: its cost, 2 x cols x k x 8 bytes, is inside kernel_ms_avg)")
            torch.cuda.empty_cache()
        except Exception as e:   # the headline line must not depend on the secondary measurement
            out["spmm"] = {"error": repr(e)[:200]}

    # ---- BASELINE configs 2 and 3 beside the headline (the driver's line must carry them) and the reference's own
    # bench shape for the SpGEMM (sprs-benches: uniform density, 2.5M x 2.5M, 4 entries per row) ------------------------------------
    if rank == 0 and world == 1 and wl == "rmat10m" and args.idx_bytes == 8 and not args.no_secondary:
        out["configs"] = {}
        for cfg in ("rmat1m", "laplace4096"):
            try:
                out["configs"][cfg] = spmv_config(dev, cfg, with_oracle=not args.no_cpu_baseline)
            except Exception as e:   # the headline line must not depend on a secondary measurement
                out["configs"][cfg] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
        try:
            out["sparse_dense_products"] = sparse_dense_micro(dev, with_oracle=not args.no_cpu_baseline)
        except Exception as e:
            out["sparse_dense_products"] = {"error": repr(e)[:200]}
        try:
            out["spgemm_uniform"] = spgemm_uniform(dev, with_cpu=not args.no_cpu_baseline)
        except Exception as e:
            out["spgemm_uniform"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()

    # ---- BASELINE config 5 beside the headline (rank 0, N = 1, default workload): a short SpGEMM object -----------
    if rank == 0 and world == 1 and wl == "rmat10m" and not args.no_secondary and not args.no_cpu_baseline:
        try:
            del sh
            handles.clear()
            torch.cuda.empty_cache()
            sg = spgemm5(dev, 8, steps=2, warmup=1, check_rows=60, cpu_blocks=8, cpu_block_rows=1000)
            out["spgemm5"] = {"seconds_per_product": sg["value"], "gflops": sg["gflops"], "nnz_c": sg["config"]["nnz_c"],
                              "roofline_frac": sg["roofline"]["frac"], "compulsory_bytes": sg["roofline"]["algorithmic_bytes_per_launch"],
                              "traffic": sg["roofline"]["traffic"], "parity": sg["parity"], "cpu_baseline": sg["cpu_baseline"],
                              "unordered_adds": sg.get("unordered_adds"),
                              "note": "python bench.py --workload spgemm5 prints the full line"}
        except Exception as e:   # the headline line must not depend on the secondary measurement
            out["spgemm5"] = {"error": repr(e)[:200]}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()                                  # no rank frees its receive window while a peer may still store into it
        peerdist = libdist = None
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
