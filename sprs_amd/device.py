"""Device containers: Python twins of the Rust wrapper crate's `DeviceCsMat` /
`DeviceVec` (rust/sprs-hip/src/lib.rs), themselves device twins of sprs'
CsMatBase (sprs/src/sparse.rs:94-122) and of the dense vectors accepted through
DenseVector (sprs/src/dense_vector.rs:10-29)."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import CSC, CSR, check, lib

_DT = {2: np.uint16, 4: np.uint32, 8: np.uint64}


def _vp(a):
    return C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)


class DeviceVec:
    """A dense f64 vector in HBM.  Either owns its buffer or borrows one
    (e.g. a torch tensor's `data_ptr()`), like a `&mut [f64]` would."""

    def __init__(self, n, ptr=None, owner=None):
        self.n = int(n)
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            check(lib.sprs_hip_malloc(C.byref(p), self.n * 8))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        v = cls(arr.size)
        check(lib.sprs_hip_memcpy_h2d(C.c_void_p(v.ptr), _vp(arr), arr.size * 8))
        return v

    @classmethod
    def zeros(cls, n):
        v = cls(n)
        check(lib.sprs_hip_memset(C.c_void_p(v.ptr), 0, v.n * 8, None))
        return v

    @classmethod
    def borrow(cls, tensor):
        """Borrow a contiguous float64 torch CUDA tensor (no copy)."""
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.element_size() == 8
        return cls(tensor.numel(), ptr=tensor.data_ptr(), owner=tensor)

    def to_host(self):
        out = np.empty(self.n, dtype=np.float64)
        check(lib.sprs_hip_synchronize(None))
        check(lib.sprs_hip_memcpy_d2h(_vp(out), C.c_void_p(self.ptr), self.n * 8))
        return out

    def dim(self):   # DenseVector::dim, dense_vector.rs:14
        return self.n

    def __len__(self):
        return self.n

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            lib.sprs_hip_free(C.c_void_p(self.ptr))
            self.ptr = 0


class DeviceCsMat:
    """A CSR/CSC matrix whose indptr / indices / data live in HBM behind a
    `sprs_hip_csmat*` handle."""

    def __init__(self, handle, keep=None):
        self._h = C.c_void_p(handle)
        self._keep = keep   # objects the handle borrows from

    # -- constructors -------------------------------------------------------
    @classmethod
    def from_host(cls, shape, indptr, indices, data, storage=CSR, validate=True):
        """CsMat::new / new_csc when validate (csmat.rs:146-166), new_trusted otherwise."""
        indptr = np.ascontiguousarray(indptr)
        indices = np.ascontiguousarray(indices)
        data = np.ascontiguousarray(data, dtype=np.float64)
        if indptr.dtype.kind not in "iu" or indices.dtype.kind not in "iu":
            raise TypeError("integer index arrays expected")
        if indices.size != data.size:
            raise _ffi.SprsHipError(_ffi.BAD_STRUCTURE, "indices and data lengths differ")
        outer = shape[0] if storage == CSR else shape[1]
        if indptr.size != outer + 1:
            raise _ffi.SprsHipError(_ffi.BAD_STRUCTURE, "Indptr length does not match dimension")
        if validate and int(indptr[-1]) - int(indptr[0]) != indices.size:
            raise _ffi.SprsHipError(_ffi.BAD_STRUCTURE, "Indices length and inpdtr's nnz do not match")
        h = C.c_void_p()
        check(lib.sprs_hip_csmat_upload(C.byref(h), storage, shape[0], shape[1], _vp(indptr),
                                        indptr.dtype.itemsize, _vp(indices), indices.dtype.itemsize,
                                        _vp(data), 1 if validate else 0))
        return cls(h.value)

    @classmethod
    def wrap_torch(cls, shape, indptr_t, indices_t, data_t, storage=CSR):
        """Borrow torch CUDA tensors as the three CSR arrays (no copy)."""
        h = C.c_void_p()
        check(lib.sprs_hip_csmat_wrap_device(
            C.byref(h), storage, shape[0], shape[1], indices_t.numel(),
            C.c_void_p(indptr_t.data_ptr()), indptr_t.element_size(),
            C.c_void_p(indices_t.data_ptr() if indices_t.numel() else 0), indices_t.element_size(),
            C.c_void_p(data_t.data_ptr() if data_t.numel() else 0)))
        return cls(h.value, keep=(indptr_t, indices_t, data_t))

    @classmethod
    def eye(cls, dim, idx_dtype=np.uint64, ptr_dtype=np.uint64):
        """CsMatI::eye (csmat.rs:416-426)."""
        return cls.from_host((dim, dim), np.arange(dim + 1, dtype=ptr_dtype),
                             np.arange(dim, dtype=idx_dtype), np.ones(dim), validate=False)

    # -- accessors ------------------------------------------------------------
    def _info(self):
        r, c, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        pb, ib, st = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.sprs_hip_csmat_info(self._h, C.byref(r), C.byref(c), C.byref(n), C.byref(pb),
                                      C.byref(ib), C.byref(st)))
        return r.value, c.value, n.value, pb.value, ib.value, st.value

    def rows(self): return self._info()[0]
    def cols(self): return self._info()[1]
    def shape(self): return self._info()[:2]
    def nnz(self): return self._info()[2]
    def storage(self): return self._info()[5]
    def is_csr(self): return self.storage() == CSR
    def is_csc(self): return self.storage() == CSC
    def index_bytes(self): return self._info()[4]
    def indptr_bytes(self): return self._info()[3]

    def to_host(self):
        """-> (shape, indptr, indices, data) numpy arrays (into_raw_storage, csmat.rs:946-954)."""
        r, c, n, pb, ib, st = self._info()
        outer = r if st == CSR else c
        indptr = np.empty(outer + 1, dtype=_DT[pb])
        indices = np.empty(n, dtype=_DT[ib])
        data = np.empty(n, dtype=np.float64)
        check(lib.sprs_hip_csmat_download(self._h, _vp(indptr), _vp(indices), _vp(data)))
        return (r, c), indptr, indices, data

    def slice_outer_to_host(self, start, end):
        """Rows [start, end) as a host view: indptr NOT rebased (slicing.rs:65-89)."""
        r, c, n, pb, ib, st = self._info()
        cnt = C.c_uint64()
        check(lib.sprs_hip_csmat_download_outer(self._h, start, end, None, None, None, C.byref(cnt)))
        indptr = np.empty(end - start + 1, dtype=_DT[pb])
        indices = np.empty(cnt.value, dtype=_DT[ib])
        data = np.empty(cnt.value, dtype=np.float64)
        check(lib.sprs_hip_csmat_download_outer(self._h, start, end, _vp(indptr), _vp(indices),
                                                _vp(data), None))
        return indptr, indices, data

    def slice_outer(self, start, end):
        """slice_outer (slicing.rs:65-89) + to_proper, materialised on the device."""
        h = C.c_void_p()
        check(lib.sprs_hip_csmat_slice_outer(self._h, start, end, C.byref(h)))
        return DeviceCsMat(h.value)

    def refresh(self):
        """Drop the cached multiply plans after the wrapped device arrays were modified in place."""
        check(lib.sprs_hip_csmat_refresh(self._h))

    def prepare(self, stream=None):
        """Build the full SpMV plan now (sprs_hip_csmat_prepare): without it the re-laid-out copy plans come with the SECOND
        multiply of the handle — the first one runs on the plain tile index."""
        sp = C.c_void_p(stream.cuda_stream) if stream is not None and hasattr(stream, "cuda_stream") else C.c_void_p(stream or 0)
        check(lib.sprs_hip_csmat_prepare(self._h, sp))
        return self

    def spmv_plan_info(self):
        """-> (kind, plan_bytes): 0 none yet, 1 nnz tiles, 2 XCD-sliced copy, 3 banded copy (hot columns from LDS)."""
        kind, nbytes = C.c_int32(), C.c_uint64()
        check(lib.sprs_hip_csmat_spmv_plan_info(self._h, C.byref(kind), C.byref(nbytes)))
        return kind.value, nbytes.value

    def transpose_view(self):
        """csmat.rs:982-991: free, shares buffers."""
        h = C.c_void_p()
        check(lib.sprs_hip_csmat_transpose_view(self._h, C.byref(h)))
        return DeviceCsMat(h.value, keep=self)

    def to_other_storage(self):
        """csmat.rs:1405-1426."""
        h = C.c_void_p()
        check(lib.sprs_hip_csmat_to_other_storage(self._h, C.byref(h)))
        return DeviceCsMat(h.value)

    # -- operators: `&A * &x`, `&A * &B` ---------------------------------------
    def __mul__(self, rhs):
        from . import prod
        if isinstance(rhs, DeviceVec):
            return prod.csmat_mul_vec(self, rhs)          # csmat.rs:2119-2160
        if isinstance(rhs, DeviceCsMat):
            return prod.csmat_mul_csmat(self, rhs)        # csmat.rs:1866-1949
        if isinstance(rhs, prod.DeviceMat):
            return prod.csmat_mul_dense(self, rhs)        # csmat.rs:1989-2048
        return NotImplemented

    __matmul__ = __mul__

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            lib.sprs_hip_csmat_free(h)
            self._h = C.c_void_p(0)
