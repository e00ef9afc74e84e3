"""Shared helpers for the parity tests."""
import numpy as np


def rel_err(got, ref):
    """max over stored values of |got-ref| / max(|got|,|ref|)  (0 where both are 0)."""
    denom = np.maximum(np.abs(got), np.abs(ref))
    out = np.zeros_like(ref, dtype=np.float64)
    nz = denom > 0
    out[nz] = np.abs(got - ref)[nz] / denom[nz]
    return float(out.max()) if out.size else 0.0


def ragged_csr(row_lengths, cols, seed=0, idx=np.uint64, ptr=np.uint64, positive=True):
    """CSR matrix with the given row lengths, sorted distinct random columns."""
    rng = np.random.default_rng(seed)
    indptr = np.zeros(len(row_lengths) + 1, dtype=np.int64)
    np.cumsum(row_lengths, out=indptr[1:])
    indices = np.empty(indptr[-1], dtype=np.int64)
    for r, n in enumerate(row_lengths):
        if n:
            assert n <= cols
            if n * 4 > cols:
                c = rng.permutation(cols)[:n]
            else:
                c = np.unique(rng.integers(0, cols, size=int(n * 1.3) + 8))
                while c.size < n:
                    c = np.unique(np.concatenate([c, rng.integers(0, cols, size=n)]))
                c = rng.permutation(c)[:n]
            indices[indptr[r]:indptr[r + 1]] = np.sort(c)
    data = rng.random(indptr[-1]) + 0.5 if positive else rng.standard_normal(indptr[-1])
    return (len(row_lengths), cols), indptr.astype(ptr), indices.astype(idx), data
