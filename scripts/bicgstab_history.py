"""Residual history of the device BiCGSTAB next to the oracle's, for a seed the fuzzer flagged: the solver is run with max_iter = 1..N
(each run is deterministic, so the runs are the prefixes of one history) and the true residual of the returned x is printed for both.
usage: python scripts/bicgstab_history.py <seed> [<seed> ...]        (test infrastructure: uses oracle/)"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity as fz                                      # noqa: E402
from sprs_amd.device import DeviceCsMat, DeviceVec
from sprs_amd.linalg import BiCGSTAB


def main():
    for seed in map(int, sys.argv[1:]):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(2, 1500))
        shape, ip, ix, dt = fz.random_csr(rng, n, n, np.uint64, np.uint64)
        m = sp.csr_matrix((dt / (1.0 + np.abs(dt)), ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
        m = (m + sp.diags(np.abs(m).sum(axis=1).A1 + 1.0)).tocsr()
        m.sort_indices()
        ip, ix, dt = m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data
        b, x0 = rng.standard_normal(n), rng.standard_normal(n)
        tol, cap = 10.0 ** -int(rng.integers(6, 13)), int(rng.integers(1, 60))
        a = DeviceCsMat.from_host(shape, ip, ix, dt)
        y = (a * DeviceVec.from_host(x0)).to_host()
        yo = np.zeros(n)
        fz.oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, x0, yo)
        spmv_bits_differ = int((y != yo).sum())
        hist = []
        for it in range(1, cap + 1):
            ref, info = fz.oracle.bicgstab(shape, ip, ix, dt, x0, b, tol, it)
            res = BiCGSTAB.solve(a, DeviceVec.from_host(x0), DeviceVec.from_host(b), tol, it)
            x = res.x().to_host()
            hist.append([it, res.iteration_count(), float(np.linalg.norm(b - m @ x)), info["iteration_count"], float(np.linalg.norm(b - m @ ref)),
                         int((x != ref).sum())])
        print(json.dumps(dict(seed=seed, n=n, nnz=int(m.nnz), tol=tol, cap=cap, norm_b=float(np.linalg.norm(b)), spmv_rows_differing=spmv_bits_differ,
                              columns=["max_iter", "gpu_iters", "gpu_resid", "ref_iters", "ref_resid", "x_entries_differing"], history=hist)))


if __name__ == "__main__":
    main()
