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
    """prod::mul_acc_mat_vec_csc (prod.rs:74-99): res_vec += mat * in_vec for a CSC matrix (sprs_hip_mul_acc_mat_vec_csc_f64:
    the matrix is converted once to CSR on the device, the copy is cached in the HANDLE, the CSR kernel runs — every result
    element still receives its products by ascending column)."""
    check(lib.sprs_hip_mul_acc_mat_vec_csc_f64(mat._h, C.c_void_p(in_vec.ptr), in_vec.n, C.c_void_p(res_vec.ptr), res_vec.n,
                                               _stream_ptr(stream)))


class DeviceMat:
    """Dense f64 matrix in HBM in one of ndarray's two contiguous layouts: `Array2` in standard (row-major) order, or
    `Array::zeros(shape.f())` (column-major) — what `&CsMat * &Array2` returns for fewer than 8 columns (csmat.rs:2017-2024)."""

    def __init__(self, rows, cols, vec=None, col_major=False):
        self.rows, self.cols, self.col_major = int(rows), int(cols), bool(col_major)
        self.vec = vec if vec is not None else DeviceVec.zeros(self.rows * self.cols)
        assert self.vec.n == self.rows * self.cols

    @classmethod
    def from_host(cls, arr, col_major=False):
        import numpy as np
        arr = np.asarray(arr, dtype=np.float64)
        flat = np.asfortranarray(arr).reshape(-1, order="F") if col_major else np.ascontiguousarray(arr).reshape(-1)
        return cls(arr.shape[0], arr.shape[1], DeviceVec.from_host(np.ascontiguousarray(flat)), col_major)

    def to_host(self):
        flat = self.vec.to_host()
        return flat.reshape(self.cols, self.rows).T if self.col_major else flat.reshape(self.rows, self.cols)

    @property
    def layout(self):
        return _ffi.COL_MAJOR if self.col_major else _ffi.ROW_MAJOR

    @property
    def ld(self):
        return self.rows if self.col_major else self.cols


def _dims(lhs, rhs, out):
    """the three dimension asserts of the four dense kernels come FIRST, the storage assert last (prod.rs:199-202, 228-231,
    256-259, 284-288): with both wrong every mirror reports the dimensions"""
    if rhs.cols != out.cols or lhs.cols() != rhs.rows or lhs.rows() != out.rows:
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")


def _mulacc_dense(lhs, rhs, out, stream):
    check(lib.sprs_hip_csmat_mulacc_dense_f64(lhs._h, C.c_void_p(rhs.vec.ptr), rhs.rows, rhs.cols, rhs.layout, rhs.ld,
                                              C.c_void_p(out.vec.ptr), out.rows, out.layout, out.ld, 1, _stream_ptr(stream)))


def _storage(lhs, want_csr):
    if lhs.is_csc() == want_csr:
        raise _ffi.SprsHipError(_ffi.STORAGE_MISMATCH, "Storage mismatch")       # prod.rs:202, 231, 258, 288


def csr_mulacc_dense_rowmaj(lhs, rhs, out, stream=None):
    """prod::csr_mulacc_dense_rowmaj (prod.rs:189-214): out += lhs * rhs."""
    _dims(lhs, rhs, out)
    _storage(lhs, True)
    _mulacc_dense(lhs, rhs, out, stream)


def csr_mulacc_dense_colmaj(lhs, rhs, out, stream=None):
    """prod::csr_mulacc_dense_colmaj (prod.rs:274-298): out += lhs * rhs (the reference walks the rhs column by column; the
    device entry is told by the operands' layouts how to address them).  rhs / out: DeviceMat, or two equally long lists of
    column vectors (out[:, j] += lhs * rhs[:, j])."""
    if isinstance(rhs, (list, tuple)):
        if len(rhs) != len(out) or any(lhs.cols() != r.n or lhs.rows() != o.n for r, o in zip(rhs, out)):
            raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")
        _storage(lhs, True)
        for r, o in zip(rhs, out):
            mul_acc_mat_vec_csr(lhs, r, o, stream)
        return
    _dims(lhs, rhs, out)
    _storage(lhs, True)
    _mulacc_dense(lhs, rhs, out, stream)


def csc_mulacc_dense_rowmaj(lhs, rhs, out, stream=None):
    """prod::csc_mulacc_dense_rowmaj (prod.rs:219-241): out += lhs * rhs for a CSC lhs.  The reference walks the columns of
    lhs in order and adds `lval * rhs[col, :]` into out[row, :], so every out[i, j] receives its products by ascending column —
    the order the CSR kernel uses on the converted matrix (cached in the handle, below the C ABI)."""
    _dims(lhs, rhs, out)
    _storage(lhs, False)
    _mulacc_dense(lhs, rhs, out, stream)


def csc_mulacc_dense_colmaj(lhs, rhs, out, stream=None):
    """prod::csc_mulacc_dense_colmaj (prod.rs:246-270); rhs / out as in csr_mulacc_dense_colmaj."""
    if isinstance(rhs, (list, tuple)):
        if len(rhs) != len(out) or any(lhs.cols() != r.n or lhs.rows() != o.n for r, o in zip(rhs, out)):
            raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")      # prod.rs:256-257
        _storage(lhs, False)
        for r, o in zip(rhs, out):
            mul_acc_mat_vec_csc(lhs, r, o, stream)
        return
    _dims(lhs, rhs, out)
    _storage(lhs, False)
    _mulacc_dense(lhs, rhs, out, stream)


def csmat_mul_dense(mat, rhs, stream=None):
    """`&CsMat * &Array2` (csmat.rs:1989-2048), ONE call below the C ABI (sprs_hip_csmat_mul_dense_f64): the four arms
    (CSR | CSC) x (>= 8 columns | fewer); the result comes back row-major for >= 8 columns and column-major (`.f()`) below,
    like the reference's."""
    out = DeviceMat(mat.rows(), rhs.cols, DeviceVec(mat.rows() * rhs.cols))
    lay = C.c_int32(0)
    check(lib.sprs_hip_csmat_mul_dense_f64(mat._h, C.c_void_p(rhs.vec.ptr), rhs.rows, rhs.cols, rhs.layout, rhs.ld,
                                           C.c_void_p(out.vec.ptr), C.byref(lay), _stream_ptr(stream)))
    out.col_major = lay.value == _ffi.COL_MAJOR
    return out


def dense_dot_csmat(lhs, mat, stream=None):
    """`Array2::dot(&CsMat)` (csmat.rs:2050-2117): dense . sparse through the transposes of the reference, below the C ABI."""
    out = DeviceMat(lhs.rows, mat.cols(), DeviceVec(lhs.rows * mat.cols()))
    lay = C.c_int32(0)
    check(lib.sprs_hip_dense_dot_csmat_f64(C.c_void_p(lhs.vec.ptr), lhs.rows, lhs.cols, lhs.layout, lhs.ld, mat._h,
                                           C.c_void_p(out.vec.ptr), C.byref(lay), _stream_ptr(stream)))
    out.col_major = lay.value == _ffi.COL_MAJOR
    return out


def csmat_mul_vec(mat, vec, out=None, stream=None):
    """`&CsMat * &Array1` (csmat.rs:2119-2160): fresh zero result; CSR goes through csr_mulacc_dense_colmaj with one column,
    CSC through csc_mulacc_dense_colmaj (csmat.rs:2140-2156) — dispatched below the C ABI (sprs_hip_csmat_mul_vec_f64)."""
    if out is None:
        out = DeviceVec(mat.rows())
    check(lib.sprs_hip_csmat_mul_vec_f64(mat._h, C.c_void_p(vec.ptr), vec.n, C.c_void_p(out.ptr), out.n, _stream_ptr(stream)))
    return out


def csmat_mul_csmat(lhs, rhs):
    """csmat_mul_csmat (csmat.rs:1895-1949): storage dispatch around smmp::mul_csr_csr, done below the C ABI
    (sprs_hip_csmat_mul_csmat); the result has the lhs' storage order."""
    from .device import DeviceCsMat
    h = C.c_void_p()
    check(lib.sprs_hip_csmat_mul_csmat(lhs._h, rhs._h, C.byref(h)))
    return DeviceCsMat(h.value)
