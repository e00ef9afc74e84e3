"""Twin of `sprs::prod` for device operands (sprs/src/sparse/prod.rs) plus the
operator dispatch of csmat.rs.  Same names, argument order and failure
behaviour: where the reference panics, SprsHipError carries the panic text."""
import ctypes as C

from . import _ffi
from ._ffi import CSC, CSR, check, lib
from .device import DeviceCsMat, DeviceVec


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))


def mul_acc_mat_vec_csr(mat, in_vec, res_vec, stream=None):
    """prod::mul_acc_mat_vec_csr (prod.rs:103-127): res_vec += mat * in_vec."""
    check(lib.sprs_hip_spmv_f64(mat._h, C.c_void_p(in_vec.ptr), in_vec.n, C.c_void_p(res_vec.ptr),
                                res_vec.n, 1, _stream_ptr(stream)))


def mul_acc_mat_vec_csc(mat, in_vec, res_vec, stream=None):
    """prod::mul_acc_mat_vec_csc (prod.rs:74-99): res_vec += mat * in_vec for a CSC matrix.
    The reference scatters column by column; on the device the matrix is converted once to CSR
    (to_other_storage, csmat.rs:1405-1426 — cached on the Python object) and the CSR kernel
    runs: same result up to the summation order inside a row."""
    if not mat.is_csc():
        raise _ffi.SprsHipError(_ffi.STORAGE_MISMATCH, "Storage mismatch")       # prod.rs:92
    if mat.cols() != in_vec.n or mat.rows() != res_vec.n:
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")          # prod.rs:88-91
    mul_acc_mat_vec_csr(_csr_of(mat), in_vec, res_vec, stream)


def csr_mulacc_dense_colmaj(lhs, rhs_cols, out_cols, stream=None):
    """prod::csr_mulacc_dense_colmaj (prod.rs:274-298) with the rhs / out given
    as lists of column vectors: out[:, j] += lhs * rhs[:, j]."""
    if len(rhs_cols) != len(out_cols):
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")
    for r, o in zip(rhs_cols, out_cols):
        mul_acc_mat_vec_csr(lhs, r, o, stream)


class DeviceMat:
    """Dense row-major f64 matrix in HBM (what `Array2<f64>` in standard layout is on the host)."""

    def __init__(self, rows, cols, vec=None):
        self.rows, self.cols = int(rows), int(cols)
        self.vec = vec if vec is not None else DeviceVec.zeros(self.rows * self.cols)
        assert self.vec.n == self.rows * self.cols

    @classmethod
    def from_host(cls, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        return cls(arr.shape[0], arr.shape[1], DeviceVec.from_host(arr.reshape(-1)))

    def to_host(self):
        return self.vec.to_host().reshape(self.rows, self.cols)


def csr_mulacc_dense_rowmaj(lhs, rhs, out, stream=None):
    """prod::csr_mulacc_dense_rowmaj (prod.rs:189-214): out += lhs * rhs, rhs/out dense row-major."""
    if rhs.cols != out.cols:
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")          # prod.rs:201
    check(lib.sprs_hip_spmm_rowmaj_f64(lhs._h, C.c_void_p(rhs.vec.ptr), rhs.rows, rhs.cols, rhs.cols,
                                       C.c_void_p(out.vec.ptr), out.rows, out.cols, 1, _stream_ptr(stream)))


def _csr_of(mat):
    """the CSR form of a CSC handle (to_other_storage on the device, csmat.rs:1405-1426), made once per Python object"""
    csr = getattr(mat, "_as_csr", None)
    if csr is None:
        csr = mat.to_other_storage()
        mat._as_csr = csr
    return csr


def csc_mulacc_dense_rowmaj(lhs, rhs, out, stream=None):
    """prod::csc_mulacc_dense_rowmaj (prod.rs:219-241): out += lhs * rhs for a CSC lhs, rhs / out dense row-major.  The
    reference walks the columns of lhs in order and adds `lval * rhs[col, :]` into out[row, :], so every out[i, j]
    receives its products by ascending column — the order the CSR kernel uses on the converted matrix."""
    if not lhs.is_csc():
        raise _ffi.SprsHipError(_ffi.STORAGE_MISMATCH, "Storage mismatch")       # prod.rs:231
    if lhs.cols() != rhs.rows or lhs.rows() != out.rows or rhs.cols != out.cols:
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")          # prod.rs:228-230
    csr_mulacc_dense_rowmaj(_csr_of(lhs), rhs, out, stream)


def csc_mulacc_dense_colmaj(lhs, rhs_cols, out_cols, stream=None):
    """prod::csc_mulacc_dense_colmaj (prod.rs:246-270) with rhs / out as lists of column vectors: out[:, j] += lhs * rhs[:, j]
    (per column the reference's scatter loop = mul_acc_mat_vec_csc)."""
    if not lhs.is_csc():
        raise _ffi.SprsHipError(_ffi.STORAGE_MISMATCH, "Storage mismatch")       # prod.rs:258
    if len(rhs_cols) != len(out_cols):
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")          # prod.rs:257
    for r, o in zip(rhs_cols, out_cols):
        mul_acc_mat_vec_csc(lhs, r, o, stream)


def csmat_mul_dense(mat, rhs, stream=None):
    """`&CsMat * &Array2` (csmat.rs:1989-2048): fresh zero result; a CSC lhs is converted once on the device; >= 8 columns use the
    row-major kernel, fewer go column by column like csr_mulacc_dense_colmaj (prod.rs:274-298) —
    here through one strided pass of the same kernel, the result stays row-major."""
    if mat.is_csc():                     # (CSC, _) arms of the dispatch, csmat.rs:2026-2045: the same product on the CSR form
        if mat.cols() != rhs.rows:
            raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")
        mat = _csr_of(mat)
    out = DeviceMat(mat.rows(), rhs.cols, DeviceVec(mat.rows() * rhs.cols))
    check(lib.sprs_hip_spmm_rowmaj_f64(mat._h, C.c_void_p(rhs.vec.ptr), rhs.rows, rhs.cols, rhs.cols,
                                       C.c_void_p(out.vec.ptr), out.rows, out.cols, 0, _stream_ptr(stream)))
    return out


def csmat_mul_vec(mat, vec, out=None, stream=None):
    """`&CsMat * &Array1` (csmat.rs:2119-2160): fresh zero result, CSR goes
    through csr_mulacc_dense_colmaj with one column."""
    if out is None:
        out = DeviceVec(mat.rows())
    if mat.is_csc():
        # csmat.rs:2149-2156 (csc_mulacc_dense_colmaj): one conversion to CSR, then the CSR kernel
        if mat.cols() != vec.n or mat.rows() != out.n:
            raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")
        check(lib.sprs_hip_memset(C.c_void_p(out.ptr), 0, out.n * 8, _stream_ptr(stream)))
        mul_acc_mat_vec_csc(mat, vec, out, stream)
        return out
    check(lib.sprs_hip_spmv_f64(mat._h, C.c_void_p(vec.ptr), vec.n, C.c_void_p(out.ptr), out.n, 0,
                                _stream_ptr(stream)))
    return out


def csmat_mul_csmat(lhs, rhs):
    """csmat_mul_csmat (csmat.rs:1895-1949): storage dispatch around smmp::mul_csr_csr, done below the C ABI
    (sprs_hip_csmat_mul_csmat); the result has the lhs' storage order."""
    from .device import DeviceCsMat
    h = C.c_void_p()
    check(lib.sprs_hip_csmat_mul_csmat(lhs._h, rhs._h, C.byref(h)))
    return DeviceCsMat(h.value)
