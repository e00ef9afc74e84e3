#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the CPU oracle: seeded random matrices (empty rows, single entries, hub rows,
unreferenced column ranges, rectangular shapes, 4- and 8-byte index types) through the SpMV plans (plain, XCD-sliced, banded
with random plan geometry), SpGEMM (every row class by random thresholds), SpMM, the Gauss-Seidel sweep, the storage dispatch of
csmat_mul_csmat, kept SpGEMM plans, triplet assembly, sliced views and BiCGSTAB, for a time budget.
Every case prints nothing when it agrees; a disagreement prints the seed and the parameters that reproduce it and counts
as a failure (exit status 1).  The oracle is the checker here, nothing else.
usage: fuzz_parity.py [seconds] [first_seed]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sprs_amd                                               # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec           # noqa: E402
from oracle import oracle                                     # noqa: E402

DEFAULTS = {}


def setopt(**kw):
    for k, v in kw.items():
        if k not in DEFAULTS:
            DEFAULTS[k] = sprs_amd.get_option(k)
        sprs_amd.set_option(k, v)


def reset():
    for k, v in DEFAULTS.items():
        sprs_amd.set_option(k, v)


def random_csr(rng, rows, cols, idx, ptr, positive=False):
    kinds = rng.integers(0, 7, size=rows)
    lens = np.select([kinds == 0, kinds == 1, kinds == 2, kinds == 3], [0, 1, rng.integers(2, 12, size=rows), rng.integers(12, 40, size=rows)],
                     rng.integers(1, 6, size=rows))
    for h in rng.choice(rows, size=min(rows, int(rng.integers(0, 4))), replace=False):
        lens[h] = rng.integers(1, cols + 1) if cols < 4000 else rng.integers(cols // 8, cols // 2)
    lens = np.minimum(lens, cols)
    ip = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=ip[1:])
    ix = np.empty(ip[-1], dtype=np.int64)
    skew = rng.random() < 0.5                                 # popular columns (power-law-ish) or uniform
    for r in range(rows):
        n = int(lens[r])
        if not n:
            continue
        if n * 3 > cols:
            c = rng.permutation(cols)[:n]
        else:
            c = np.zeros(0, dtype=np.int64)
            while c.size < n:
                draw = (cols * rng.random(2 * n + 8) ** (3.0 if skew else 1.0)).astype(np.int64)
                c = np.unique(np.concatenate([c, np.minimum(draw, cols - 1)]))
            c = rng.permutation(c)[:n]
        ix[ip[r]:ip[r + 1]] = np.sort(c)
    dt = (rng.random(ip[-1]) + 0.5) if positive else rng.standard_normal(ip[-1]) * 10.0 ** rng.integers(-3, 4, size=ip[-1])
    return (rows, cols), ip.astype(ptr), ix.astype(idx), dt


def types(rng):
    return [(np.uint64, np.uint64), (np.uint32, np.uint64), (np.uint32, np.uint32)][int(rng.integers(0, 3))]


def case_spmv(rng):
    idx, ptr = types(rng)
    rows, cols = int(rng.integers(1, 3000)), int(rng.integers(1, 40000))
    shape, ip, ix, dt = random_csr(rng, rows, cols, idx, ptr)
    plan = int(rng.integers(0, 4))
    opts = {}
    if plan == 1:
        opts = dict(spmv_band=2, spmv_xcs=1, spmv_xcs_split=int(rng.choice([2, 8, 32])))
    elif plan == 2:
        tile = int(rng.choice([8192, 16384]))
        opts = dict(spmv_band=1, spmv_band_tile=tile, spmv_band_hot=int(rng.integers(1, max(2, cols // tile + 2))),
                    spmv_band_split=int(rng.choice([2, 8, 24, 40])), spmv_band_rounds=int(rng.integers(1, 5)),
                    spmv_band_hot_run=int(rng.integers(1, 5)), spmv_band_cold_tiles=int(rng.integers(1, 5)),
                    spmv_band_phases=int(rng.integers(1, 3)),
                    # round 5: one stream / two streams, the reduction behind hot slices + cold pieces or behind the whole second stream
                    spmv_band_overlap=int(rng.choice([0, 1, 2])), spmv_band_tail=int(rng.choice([0, 2])))
    elif plan == 3:
        opts = dict(spmv_band=2, spmv_xcs=2, spmv_kernel=int(rng.choice([1, 2])), spmv_tile=int(rng.choice([0, 2048, 4096])))
    setopt(**opts)
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    x = rng.standard_normal(cols)
    y0 = rng.standard_normal(rows)
    from sprs_amd import prod
    ref = oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, x, y0.copy())
    yd = DeviceVec.from_host(y0)
    prod.mul_acc_mat_vec_csr(a, DeviceVec.from_host(x), yd)
    got = yd.to_host()
    import scipy.sparse as sp
    mag = np.abs(sp.csr_matrix((np.abs(dt), ix.astype(np.int64), ip.astype(np.int64)), shape=shape)) @ np.abs(x) + np.abs(y0)
    bad = np.abs(got - ref) > 1e-10 * np.maximum(np.abs(ref), 1e-300) + 64 * np.finfo(float).eps * mag
    plain = prod.csmat_mul_vec(a, DeviceVec.from_host(x)).to_host()
    ref2 = oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, x, np.zeros(rows))
    bad2 = np.abs(plain - ref2) > 1e-10 * np.maximum(np.abs(ref2), 1e-300) + 64 * np.finfo(float).eps * mag
    return not (bad.any() or bad2.any()), dict(kind="spmv", rows=rows, cols=cols, plan=plan, opts=opts, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)))


def case_spgemm(rng):
    idx, ptr = types(rng)
    m, k, n = int(rng.integers(1, 400)), int(rng.integers(1, 900)), int(rng.integers(1, 300000))
    A = random_csr(rng, m, k, idx, ptr)
    B = random_csr(rng, k, n, idx, ptr)
    opts = dict(spgemm_mid=int(rng.choice([0, 600, 65536])), spgemm_heavy=int(rng.choice([1024, 4096, 131072])),
                spgemm_winlog=int(rng.choice([16, 17, 18, 19])), spgemm_bucket=int(rng.integers(0, 2)),
                spgemm_tokens=int(rng.choice([1, 2, 4])),
                spgemm_lane_order=int(rng.choice([0, 2])), spgemm_keep_bits=int(rng.integers(0, 2)), spgemm_midwin=int(rng.choice([14, 15])), spgemm_mid_keep=int(rng.choice([4, 8])), spgemm_mid_keep_sym=int(rng.choice([8, 16])), spgemm_midwin_sym=int(rng.choice([14, 16])))
    setopt(**opts)
    try:
        rs, rip, rix, rdt = oracle.mul_csr_csr(A[0], A[1], A[2], A[3], B[0], B[1], B[2], B[3])
    except oracle.OracleError:
        return True, dict(kind="spgemm", skipped="index overflow in the oracle")
    c = (DeviceCsMat.from_host(*A) * DeviceCsMat.from_host(*B)).to_host()
    ok = c[0] == tuple(rs) and np.array_equal(c[1], rip) and np.array_equal(c[2], rix) and np.array_equal(c[3].view(np.uint64), rdt.view(np.uint64))
    return ok, dict(kind="spgemm", m=m, k=k, n=n, opts=opts, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)))


def case_spmm(rng):
    idx, ptr = types(rng)
    rows, cols, k = int(rng.integers(1, 1500)), int(rng.integers(1, 5000)), int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 33, 64, 70]))
    shape, ip, ix, dt = random_csr(rng, rows, cols, idx, ptr)
    rhs = rng.standard_normal((cols, k))
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    import ctypes as C
    from sprs_amd import _ffi
    # the summation modes: entry stream (from the rhs itself / from its re-laid-out copy), chunks, lane groups in entry order for rows of
    # <= L entries; the accumulate form
    mode = int(rng.integers(0, 4))
    opts = {0: {}, 1: dict(spmm_relayout=1), 2: dict(spmm_stream=0), 3: dict(spmm_long_row=int(rng.choice([1, 5, 40, 600])))}[mode]      # (1: the rhs gathered from its re-laid-out copy)
    setopt(**opts)
    acc = bool(rng.integers(0, 2))
    out0 = rng.standard_normal((rows, k)) if acc else np.zeros((rows, k))
    d_rhs, d_out = DeviceVec.from_host(rhs.reshape(-1)), DeviceVec.from_host(out0.reshape(-1).copy())
    _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(a._h, C.c_void_p(d_rhs.ptr), cols, k, k, C.c_void_p(d_out.ptr), rows, k, int(acc), None))
    got = d_out.to_host().reshape(rows, k)
    import scipy.sparse as sp
    absm = sp.csr_matrix((np.abs(dt), ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    ok = True
    for j in range(k):
        ref = oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, np.ascontiguousarray(rhs[:, j]), np.ascontiguousarray(out0[:, j]).copy())
        mag = absm @ np.abs(rhs[:, j]) + np.abs(out0[:, j])
        ok = ok and not np.any(np.abs(got[:, j] - ref) > 1e-10 * np.abs(ref) + 64 * np.finfo(float).eps * mag)
    return ok, dict(kind="spmm", rows=rows, cols=cols, k=k, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)), opts=opts, accumulate=acc)


def case_gauss_seidel(rng):
    from sprs_amd.linalg import gauss_seidel
    import scipy.sparse as sp
    idx, ptr = types(rng)
    n = int(rng.integers(1, 3000))
    shape, ip, ix, dt = random_csr(rng, n, n, np.uint64, np.uint64)
    m = sp.csr_matrix((dt, ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    m = (m + sp.diags(np.abs(m).sum(axis=1).A1 + 1.0)).tocsr()
    m.sort_indices()
    ip, ix, dt = m.indptr.astype(ptr), m.indices.astype(idx), m.data
    rhs, x0 = rng.standard_normal(n), rng.standard_normal(n)
    sweeps = int(rng.integers(1, 4))
    opts = dict(gauss_seidel_xcd=int(rng.integers(0, 3)), gauss_seidel_blocks=int(rng.choice([0, 1, 3, 64])))
    setopt(**opts)
    ref, info = oracle.gauss_seidel(shape, ip, ix, dt, x0, rhs, sweeps, -1.0)
    x = DeviceVec.from_host(x0)
    gauss_seidel(DeviceCsMat.from_host(shape, ip, ix, dt), x, DeviceVec.from_host(rhs), sweeps, -1.0)
    return bool(np.array_equal(x.to_host().view(np.uint64), ref.view(np.uint64))), dict(kind="gauss_seidel", n=n, sweeps=sweeps, opts=opts)


def case_dispatch(rng):
    """csmat_mul_csmat (csmat.rs:1895-1949): the four storage combinations below the ABI, and to_other_storage on its own"""
    from sprs_amd.device import CSR, CSC
    from sprs_amd import prod
    idx, ptr = types(rng)
    m, k, n = int(rng.integers(1, 200)), int(rng.integers(1, 300)), int(rng.integers(1, 5000))
    ls, rs = ("CSR", "CSC")[int(rng.integers(0, 2))], ("CSR", "CSC")[int(rng.integers(0, 2))]
    # a CSC operand of shape (r, c) is stored as the CSR arrays of its transpose (c x r)
    A = random_csr(rng, *((m, k) if ls == "CSR" else (k, m)), idx, ptr)
    B = random_csr(rng, *((k, n) if rs == "CSR" else (n, k)), idx, ptr)
    lhs = dict(storage=ls, shape=(m, k), indptr=A[1], indices=A[2], data=A[3])
    rhs = dict(storage=rs, shape=(k, n), indptr=B[1], indices=B[2], data=B[3])
    try:
        ref = oracle.csmat_mul_csmat(lhs, rhs)
    except oracle.OracleError:
        return True, dict(kind="dispatch", skipped="index overflow in the oracle")
    dl = DeviceCsMat.from_host((m, k), A[1], A[2], A[3], storage=CSR if ls == "CSR" else CSC)
    dr = DeviceCsMat.from_host((k, n), B[1], B[2], B[3], storage=CSR if rs == "CSR" else CSC)
    c = prod.csmat_mul_csmat(dl, dr)
    got = c.to_host()
    ok = (got[0] == tuple(ref["shape"]) and c.storage() == (CSR if ref["storage"] == "CSR" else CSC) and np.array_equal(got[1], ref["indptr"])
          and np.array_equal(got[2], ref["indices"]) and np.array_equal(got[3].view(np.uint64), ref["data"].view(np.uint64)))
    o = dl.to_other_storage().to_host()
    outer, inner = (m, k) if ls == "CSR" else (k, m)
    rip, rix, rdt = oracle.convert_storage(outer, inner, A[1], A[2], A[3])
    ok = ok and np.array_equal(o[1], rip) and np.array_equal(o[2], rix) and np.array_equal(o[3].view(np.uint64), rdt.view(np.uint64))
    return ok, dict(kind="dispatch", m=m, k=k, n=n, lhs=ls, rhs=rs, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)))


def case_kept_plan(rng):
    """smmp::symbolic once, smmp::numeric for new values of the same structure (sprs_hip_spgemm_plan_*)"""
    from sprs_amd import smmp
    idx, ptr = types(rng)
    m, k, n = int(rng.integers(1, 300)), int(rng.integers(1, 500)), int(rng.integers(1, 100000))
    A = random_csr(rng, m, k, idx, ptr)
    B = random_csr(rng, k, n, idx, ptr)
    try:
        ref = oracle.mul_csr_csr(A[0], A[1], A[2], A[3], B[0], B[1], B[2], B[3])
    except oracle.OracleError:
        return True, dict(kind="kept_plan", skipped="index overflow in the oracle")
    da, db = DeviceCsMat.from_host(*A), DeviceCsMat.from_host(*B)
    plan = smmp.SpgemmPlan(da, db)
    st = plan.structure().to_host()
    ok = plan.nnz() == ref[2].size and np.array_equal(st[1], ref[1]) and np.array_equal(st[2], ref[2]) and not np.any(st[3])
    c = plan.product()
    got = c.to_host()
    ok = ok and np.array_equal(got[3].view(np.uint64), ref[3].view(np.uint64))
    A2 = (A[0], A[1], A[2], rng.standard_normal(A[3].size))
    B2 = (B[0], B[1], B[2], rng.standard_normal(B[3].size))
    ref2 = oracle.mul_csr_csr(A2[0], A2[1], A2[2], A2[3], B2[0], B2[1], B2[2], B2[3])
    import ctypes as C
    from sprs_amd._ffi import check, lib
    for h, vals in ((da, A2[3]), (db, B2[3])):                # new VALUES in place: the plan belongs to these two handles
        p_ip, p_ix, p_dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.sprs_hip_csmat_device_ptrs(h._h, C.byref(p_ip), C.byref(p_ix), C.byref(p_dt)))
        if vals.size:
            check(lib.sprs_hip_memcpy_h2d(p_dt, vals.ctypes.data_as(C.c_void_p), vals.size * 8))
    plan.numeric(c)
    got2 = c.to_host()
    ok = ok and np.array_equal(got2[2], ref2[2]) and np.array_equal(got2[3].view(np.uint64), ref2[3].view(np.uint64))
    return ok, dict(kind="kept_plan", m=m, k=k, n=n, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)))


def case_triplets(rng):
    """TriMatBase::to_csr / to_csc (triplet_iter.rs:127-224): duplicates folded in triplet order, explicit zeros kept"""
    from sprs_amd.triplet import TriMat
    rows, cols = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    n = int(rng.integers(0, 6000))
    r = rng.integers(0, rows, size=n)
    c = rng.integers(0, cols, size=n)
    if n > 8:                                                  # plenty of duplicates
        dup = rng.integers(0, n, size=n // 3)
        r[dup], c[dup] = r[(dup * 7) % n], c[(dup * 7) % n]
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 9, size=n)
    v[rng.random(n) < 0.05] = 0.0
    ok = True
    for storage in ("CSR", "CSC"):
        ref = oracle.triplets_to_cs((rows, cols), r, c, v, storage=storage)
        t = TriMat.from_triplets((rows, cols), r, c, v)
        got = (t.to_csr() if storage == "CSR" else t.to_csc()).to_host()
        ok = ok and np.array_equal(got[1], ref[0]) and np.array_equal(got[2], ref[1]) and np.array_equal(got[3].view(np.uint64), ref[2].view(np.uint64))
    return ok, dict(kind="triplets", rows=rows, cols=cols, n=n)


def case_sliced_view(rng):
    """slice_outer views: non-zero-based indptr on upload (indptr.rs:206-219) and the device-side slice, through the SpMV"""
    from sprs_amd import prod
    idx, ptr = types(rng)
    rows, cols = int(rng.integers(2, 2000)), int(rng.integers(1, 20000))
    shape, ip, ix, dt = random_csr(rng, rows, cols, idx, ptr)
    a, b = sorted(int(v) for v in rng.integers(0, rows + 1, size=2))
    x = rng.standard_normal(cols)
    ref = oracle.mul_acc_mat_vec_csr((rows, cols), ip, ix, dt, x, np.zeros(rows))[a:b]
    s0, e0 = int(ip[a]), int(ip[b])
    view = DeviceCsMat.from_host((b - a, cols), ip[a:b + 1], ix[s0:e0], dt[s0:e0])          # indptr NOT rebased by the caller
    got1 = prod.csmat_mul_vec(view, DeviceVec.from_host(x)).to_host() if b > a else np.zeros(0)
    whole = DeviceCsMat.from_host(shape, ip, ix, dt)
    got2 = prod.csmat_mul_vec(whole.slice_outer(a, b), DeviceVec.from_host(x)).to_host() if b > a else np.zeros(0)
    import scipy.sparse as sp
    mag = (np.abs(sp.csr_matrix((np.abs(dt), ix.astype(np.int64), ip.astype(np.int64)), shape=shape)) @ np.abs(x))[a:b]
    tol = 1e-10 * np.abs(ref) + 64 * np.finfo(float).eps * mag
    ok = not (np.any(np.abs(got1 - ref) > tol) or np.any(np.abs(got2 - ref) > tol))
    return ok, dict(kind="sliced_view", rows=rows, cols=cols, a=a, b=b, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)))


def case_bicgstab(rng):
    """BiCGSTAB::solve (bicgstab.rs:148-171), serial-dot sizes: same counters and bits as the restatement"""
    from sprs_amd.linalg import BiCGSTAB
    import scipy.sparse as sp
    n = int(rng.integers(2, 1500))
    shape, ip, ix, dt = random_csr(rng, n, n, np.uint64, np.uint64)
    m = sp.csr_matrix((dt / (1.0 + np.abs(dt)), ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    m = (m + sp.diags(np.abs(m).sum(axis=1).A1 + 1.0)).tocsr()
    m.sort_indices()
    ip, ix, dt = m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    tol, it = 10.0 ** -int(rng.integers(6, 13)), int(rng.integers(1, 60))
    ref, info = oracle.bicgstab(shape, ip, ix, dt, x0, b, tol, it)
    res = BiCGSTAB.solve(DeviceCsMat.from_host(shape, ip, ix, dt), DeviceVec.from_host(x0), DeviceVec.from_host(b), tol, it)
    x = res.x().to_host()
    # BiCGSTAB amplifies the rounding of its SpMVs (a row that straddles two tiles is summed as tail + head: within the 1e-10 bar,
    # not bit-identical), so iterates of a slowly converging system drift apart — not a parity question.  What must hold:
    # the reported err is the norm of the true residual of the returned x when convergence was claimed (hard restart), the
    # two solvers reach comparable residuals, and where both converged the solutions agree to the tolerance asked for.
    r_gpu, r_ref = np.linalg.norm(b - m @ x), np.linalg.norm(b - m @ ref)
    bound = 1e3 * max(r_ref, tol)
    # ... and when the oracle converged only just inside the iteration cap while the device run is still on its way there (seed 716939:
    # 46 of 49 allowed iterations against 49 with a residual of 1.7e-9 for tol 1e-12 — the same numbers from the round-4 sources and
    # from the CPU emulator: a property of the summation order of the small-plan SpMV, not of a build)
    near_cap = bool(info["converged"]) and not res.converged and info["iteration_count"] + max(5, it // 10) >= it
    if (not res.converged and not info["converged"]) or near_cap:
        # both stopped by the iteration cap: a history branches on the soft-restart test |rho| / err^2 < 0.1 by rounding alone (seeds 57863
        # and 60527: profiles/r11b_bicgstab_history.jsonl next to the two dot orders on the CPU, profiles/r11b_bicgstab_dot_order_cpu.txt),
        # and a restarted run trails the other by orders of magnitude for the rest of it — only progress can be asked of such a run
        bound = max(bound, 1e-3 * np.linalg.norm(b - m @ x0))
    ok = bool(np.isfinite(r_gpu) and r_gpu <= bound and res.iteration_count() <= it)
    if not np.isfinite(r_ref) and not np.isfinite(r_gpu):        # a breakdown (0 / 0 in the recurrences) on both sides: the same NaN story
        ok = res.iteration_count() == info["iteration_count"] and bool(res.converged) == bool(info["converged"])
    if res.converged:
        ok = ok and abs(res.err() - r_gpu) <= 1e-6 * max(r_gpu, 1e-300) + 1e-14 * np.linalg.norm(b) and res.err() < tol
    if res.converged and info["converged"]:
        ok = ok and np.abs(x - ref).max() <= 1e3 * tol * max(1.0, np.abs(ref).max())
    return bool(ok), dict(kind="bicgstab", n=n, tol=tol, max_iter=it, gpu=(res.iteration_count(), float(r_gpu)), ref=(info["iteration_count"], float(r_ref)))


def case_dense(rng):
    """the dense-operand dispatch below the ABI: `&CsMat * &Array2` (both storages, both layouts of the rhs, the layout rule of the
    result), the four accumulate kernels with explicit layouts, `&CsMat * &Array1` on CSC, dense . sparse"""
    import scipy.sparse as sp
    from sprs_amd import prod
    from sprs_amd.device import CSC
    idx, ptr = types(rng)
    rows, cols, k = int(rng.integers(1, 900)), int(rng.integers(1, 1200)), int(rng.choice([1, 2, 5, 7, 8, 9, 17, 40]))
    shape, ip, ix, dt = random_csr(rng, rows, cols, idx, ptr)
    m = sp.csr_matrix((dt, ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    absm = abs(m)
    csc = bool(rng.integers(0, 2))
    if csc:
        mc = m.tocsc()
        mc.sort_indices()
        a = DeviceCsMat.from_host(shape, mc.indptr.astype(ptr), mc.indices.astype(idx), mc.data, storage=CSC)
    else:
        a = DeviceCsMat.from_host(shape, ip, ix, dt)
    rhs = rng.standard_normal((cols, k))
    rcm, ocm = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    setopt(spmm_relayout=int(rng.integers(0, 3)))                 # auto (a column-major rhs of >= 8 columns is re-laid-out) / always / never
    tol = lambda got, ref, mag: np.abs(got - ref) <= 1e-10 * np.maximum(np.abs(ref), 1e-300) + 256 * np.finfo(float).eps * mag
    out = a * prod.DeviceMat.from_host(rhs, col_major=rcm)
    ok = out.col_major == (k < 8) and bool(tol(out.to_host(), m @ rhs, absm @ np.abs(rhs)).all())
    out0 = rng.standard_normal((rows, k))
    acc = prod.DeviceMat.from_host(out0, col_major=ocm)
    kern = {(False, False): prod.csr_mulacc_dense_rowmaj, (False, True): prod.csr_mulacc_dense_colmaj,
            (True, False): prod.csc_mulacc_dense_rowmaj, (True, True): prod.csc_mulacc_dense_colmaj}[(csc, ocm)]
    kern(a, prod.DeviceMat.from_host(rhs, col_major=rcm), acc)
    ok = ok and bool(tol(acc.to_host(), out0 + m @ rhs, np.abs(out0) + absm @ np.abs(rhs)).all())
    x = rng.standard_normal(cols)
    ok = ok and bool(tol((a * DeviceVec.from_host(x)).to_host(), m @ x, absm @ np.abs(x)).all())
    lhs = rng.standard_normal((int(rng.choice([1, 3, 9])), rows))
    got = prod.dense_dot_csmat(prod.DeviceMat.from_host(lhs, col_major=rcm), a).to_host()
    ok = ok and bool(tol(got, lhs @ m.toarray(), np.abs(lhs) @ absm.toarray()).all())
    return ok, dict(kind="dense", rows=rows, cols=cols, k=k, csc=csc, rhs_col_major=rcm, out_col_major=ocm, idx=str(np.dtype(idx)), ptr=str(np.dtype(ptr)))


CASES = (case_dense, case_spmv, case_spgemm, case_spmm, case_gauss_seidel, case_dispatch, case_kept_plan, case_triplets, case_sliced_view,
         case_spmv, case_spgemm, case_bicgstab)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = os.environ.get("FUZZ_KINDS")                       # e.g. FUZZ_KINDS=spgemm,kept_plan
    cases = [c for c in CASES if not only or c.__name__[5:] in only.split(",")]
    t0, counts, failures = time.time(), {}, []
    while time.time() - t0 < seconds:
        rng = np.random.default_rng(seed)
        fn = cases[seed % len(cases)]
        try:
            ok, what = fn(rng)
        except Exception as e:                                # a status the oracle does not mirror is a finding too
            ok, what = False, dict(kind=fn.__name__, error=repr(e)[:300])
        finally:
            reset()
        counts[what.get("kind", "?")] = counts.get(what.get("kind", "?"), 0) + 1
        if not ok:
            failures.append(dict(seed=seed, **what))
            print(json.dumps(failures[-1]), flush=True)
        seed += 1
    print(json.dumps({"cases": counts, "failures": len(failures), "seconds": round(time.time() - t0, 1), "next_seed": seed}))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
