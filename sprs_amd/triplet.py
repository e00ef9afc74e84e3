"""Triplet (COO) matrices and their conversion to CSR / CSC on the device — twin of
`TriMatBase::to_csr / to_csc` (sprs/src/sparse/triplet.rs:262-276 -> TriMatIter::into_cs,
sprs/src/sparse/triplet_iter.rs:127-224): entries sorted by (outer, inner), DUPLICATES SUMMED.

Default route: `sprs_hip_triplets_to_cs` (sprs_amd/csrc/triplet.hip): keys (outer << 32 | inner), a stable device
radix sort (sort.hip), duplicates folded left to right in triplet order, indptr filled including empty slices.

Alternative route (method="product", the first implementation, kept as a cross-check): no kernel of its own.
With n triplets (r_p, c_p, v_p), let
    R  (rows x n):  R[r_p, p] = 1      — its CSC arrays are (0..n, r, ones): one entry per column
    E  (n x cols):  E[p, c_p] = v_p    — its CSR arrays are (0..n, c, v):    one entry per row
then  A = R * E  is the assembled matrix: A[i, j] = sum over the triplets p with (r_p, c_p) = (i, j) of
1 * v_p, and the SpGEMM adds over k = p ASCENDING (smmp.rs:174-181).  So the device path is: upload the
two trivially valid matrices, one device `to_other_storage` (the counting-sort conversion of
csmat.rs:1782-1829) and one `smmp::mul_csr_csr` — rows come out sorted, duplicates are summed in
triplet order, explicit zeros and cancelled sums stay stored, exactly as `into_cs` keeps them.

Order of the duplicate sums: the reference sorts with `sort_unstable_by_key`, i.e. it leaves the
order of equal (row, col) keys unspecified; triplet order is one of its possible outcomes and the one
a stable sort gives.  Two duplicates commute, so the result differs from ANY outcome of the reference
only for cells with three or more entries, and then only in rounding.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import CSC, CSR, check, lib
from .device import DeviceCsMat, DeviceVec


class TriMat:
    """Host triplet matrix (TriMatI<f64, I>, triplet.rs:26-48)."""

    def __init__(self, shape, row_inds=(), col_inds=(), data=()):
        self.rows, self.cols = int(shape[0]), int(shape[1])
        self.row_inds = np.ascontiguousarray(row_inds, dtype=np.uint64)
        self.col_inds = np.ascontiguousarray(col_inds, dtype=np.uint64)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        # TriMatI::from_triplets asserts (triplet.rs:96-113)
        if not (self.row_inds.size == self.col_inds.size == self.data.size):
            raise ValueError("all inputs should have the same length")
        if self.row_inds.size and (int(self.row_inds.max()) >= self.rows or int(self.col_inds.max()) >= self.cols):
            raise ValueError("index out of bounds")

    @classmethod
    def from_triplets(cls, shape, row_inds, col_inds, data):
        return cls(shape, row_inds, col_inds, data)

    def shape(self):
        return self.rows, self.cols

    def nnz(self):
        """number of stored triplets, duplicates included (triplet.rs:183-185)"""
        return int(self.data.size)

    def add_triplet(self, row, col, val):
        if not (0 <= row < self.rows and 0 <= col < self.cols):
            raise ValueError("index out of bounds")
        self.row_inds = np.append(self.row_inds, np.uint64(row))
        self.col_inds = np.append(self.col_inds, np.uint64(col))
        self.data = np.append(self.data, float(val))

    # ---- device conversions ---------------------------------------------------------------------
    def _selectors(self, idx_dtype):
        n = self.nnz()
        if max(self.rows, self.cols, n) >= 2 ** 32 - 1 and np.dtype(idx_dtype).itemsize == 4:
            raise OverflowError("Index type is not large enough to hold the matrix")     # SpIndex::from_usize
        ptr = np.arange(n + 1, dtype=np.uint64)
        return n, ptr, self.row_inds.astype(idx_dtype), self.col_inds.astype(idx_dtype)

    def _assemble(self, storage, idx_dtype):
        n = self.nnz()
        ib = np.dtype(idx_dtype).itemsize
        if max(self.rows, self.cols) >= 2 ** 32 - 1 and ib == 4:
            raise OverflowError("Index type is not large enough to hold the matrix")     # SpIndex::from_usize
        bufs = []

        def up(arr):
            p = C.c_void_p()
            check(lib.sprs_hip_malloc(C.byref(p), max(arr.nbytes, 8)))
            bufs.append(p)
            if arr.nbytes:
                check(lib.sprs_hip_memcpy_h2d(p, C.c_void_p(arr.ctypes.data), arr.nbytes))
            return p

        try:
            r, c, v = up(self.row_inds), up(self.col_inds), up(self.data)
            h = C.c_void_p()
            check(lib.sprs_hip_triplets_to_cs(self.rows, self.cols, n, r, c, 8, v, storage, ib, 8, C.byref(h)))
            return DeviceCsMat(h.value)
        finally:
            for p in bufs:
                lib.sprs_hip_free(p)

    def to_csr(self, idx_dtype=np.uint64, method="sort"):
        """TriMatBase::to_csr (triplet.rs:270-276).  Index type I = idx_dtype, Iptr = u64."""
        if method == "sort":
            return self._assemble(CSR, idx_dtype)
        from . import smmp
        n, ptr, r, c = self._selectors(idx_dtype)
        if n == 0:
            return DeviceCsMat.from_host((self.rows, self.cols), np.zeros(self.rows + 1, dtype=np.uint64),
                                         np.zeros(0, dtype=idx_dtype), np.zeros(0), validate=False)
        sel = DeviceCsMat.from_host((self.rows, n), ptr, r, np.ones(n), storage=CSC).to_other_storage()   # R as CSR
        ent = DeviceCsMat.from_host((n, self.cols), ptr, c, self.data)                                   # E
        return smmp.mul_csr_csr(sel, ent)

    def to_csc(self, idx_dtype=np.uint64, method="sort"):
        """TriMatBase::to_csc (triplet.rs:262-268); method="product": computed as (E^T R^T)^T, whose CSR arrays are A's CSC arrays."""
        if method == "sort":
            return self._assemble(CSC, idx_dtype)
        from . import smmp
        n, ptr, r, c = self._selectors(idx_dtype)
        if n == 0:
            return DeviceCsMat.from_host((self.rows, self.cols), np.zeros(self.cols + 1, dtype=np.uint64),
                                         np.zeros(0, dtype=idx_dtype), np.zeros(0), storage=CSC, validate=False)
        ent_t = DeviceCsMat.from_host((n, self.cols), ptr, c, self.data).to_other_storage().transpose_view()   # cols x n
        sel_t = DeviceCsMat.from_host((n, self.rows), ptr, r, np.ones(n))                                      # n x rows
        return smmp.mul_csr_csr(ent_t, sel_t).transpose_view()
