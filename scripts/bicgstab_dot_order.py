"""Two CPU runs of the BiCGSTAB recurrence (bicgstab.rs:148-229 as restated in oracle/sprs_oracle_impl.h) on a fuzz seed, differing only in the
summation order of the dot products (serial like the reference / numpy pairwise): shows where a history branches on the soft-restart test
|rho| / err^2 < 0.1 by rounding alone.  usage: python scripts/bicgstab_dot_order.py   (test infrastructure)"""
import sys, numpy as np, scipy.sparse as sp
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity as fz
def gen(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 1500))
    shape, ip, ix, dt = fz.random_csr(rng, n, n, np.uint64, np.uint64)
    m = sp.csr_matrix((dt / (1.0 + np.abs(dt)), ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    m = (m + sp.diags(np.abs(m).sum(axis=1).A1 + 1.0)).tocsr(); m.sort_indices()
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    tol, it = 10.0 ** -int(rng.integers(6, 13)), int(rng.integers(1, 60))
    return m, b, x0, tol, it
def solve(m, b, x0, tol, cap, dot):
    x = x0.copy(); r = b - m @ x; rhat = r.copy(); p = r.copy()
    err = np.sqrt(dot(r, r)); rho = err * err; log = []
    for k in range(cap):
        v = m @ p; alpha = rho / dot(rhat, v); h = x + p * alpha; s = r - v * alpha
        t = m @ s; omega = dot(t, s) / dot(t, t); x = h + omega * s; r = s - t * omega
        err = np.sqrt(dot(r, r)); rho_prev = rho; rho = dot(rhat, r)
        ratio = abs(rho) / (err * err); soft = ratio < 0.1
        if soft: rhat = r.copy(); p = r.copy(); rho = err * err
        else: p = r + (p - v * omega) * ((rho / rho_prev) * (alpha / omega))
        hard = err < tol
        if hard:
            r = b - m @ x; err = np.sqrt(dot(r, r)); rhat = r.copy(); p = r.copy(); rho = err * err
            if err < tol: break
        log.append((k + 1, float(np.linalg.norm(b - m @ x)), float(ratio), bool(soft), bool(hard)))
    return log
serial = lambda a, b: float(np.cumsum(a * b)[-1]) if a.size else 0.0     # serial order like the reference
for seed in (57863, 60527):
    m, b, x0, tol, cap = gen(seed)
    la = solve(m, b, x0, tol, cap, serial); lb = solve(m, b, x0, tol, cap, lambda a, b: float(np.dot(a, b)))
    print(seed)
    for a, c in zip(la, lb):
        print("  %2d serial %.3e ratio %.4f %s%s | pairwise %.3e ratio %.4f %s%s" % (a[0], a[1], a[2], "S" if a[3] else "-", "H" if a[4] else "-", c[1], c[2], "S" if c[3] else "-", "H" if c[4] else "-"))
