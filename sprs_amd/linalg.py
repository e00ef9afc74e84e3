"""Host mirror of the reference's two callers that loop on the SpMV, over the device path:
sprs::linalg::bicgstab (sprs/src/sparse/linalg/bicgstab.rs) behind `sprs_hip_bicgstab_f64` (every vector resident
in HBM: two SpMVs, four fused element-wise kernels and three dot launches per iteration) and the Gauss-Seidel
solver of the heat example (sprs/examples/heat.rs:103-139) behind `sprs_hip_gauss_seidel_f64`."""
import ctypes as C

from . import _ffi
from ._ffi import check, lib
from .device import DeviceVec


class _Info(C.Structure):
    _fields_ = [("iteration_count", C.c_uint64), ("soft_restart_count", C.c_uint64),
                ("hard_restart_count", C.c_uint64), ("err", C.c_double), ("rho", C.c_double),
                ("converged", C.c_int32)]


class BiCGSTAB:
    """Result of a solve, with the reference's accessors (bicgstab.rs:236-300).  The reference returns
    `Ok(solver)` / `Err(solver)` (iteration limit reached, results still inside); here `converged`
    tells the two apart and `solve` raises nothing for the latter, as `Err` is not a panic."""

    def __init__(self, a, x, b, info, soft_restart_threshold):
        self._a, self._x, self._b, self._i, self._thr = a, x, b, info, soft_restart_threshold

    @classmethod
    def solve(cls, a, x0, b, tol, max_iter, soft_restart_threshold=0.1, stream=None):
        """BiCGSTAB::solve(a, x0, b, tol, max_iter) (bicgstab.rs:148-171).  a: square DeviceCsMat (CSR or
        CSC), x0 / b: DeviceVec.  Dimension mismatch raises like the reference's panics."""
        n = x0.n
        if b.n != n:
            raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")
        x = DeviceVec(n)
        info = _Info()
        check(lib.sprs_hip_bicgstab_f64(a._h, C.c_void_p(x0.ptr), C.c_void_p(b.ptr), n, float(tol), int(max_iter),
                                        float(soft_restart_threshold), C.c_void_p(x.ptr), C.byref(info),
                                        C.c_void_p(int(stream) if stream else 0)))
        return cls(a, x, b, info, soft_restart_threshold)

    @property
    def converged(self):
        return bool(self._i.converged)

    def iteration_count(self):
        return int(self._i.iteration_count)

    def soft_restart_count(self):
        return int(self._i.soft_restart_count)

    def hard_restart_count(self):
        return int(self._i.hard_restart_count)

    def soft_restart_threshold(self):
        return self._thr

    def err(self):
        return float(self._i.err)

    def rho(self):
        return float(self._i.rho)

    def a(self):
        return self._a

    def x(self):
        return self._x

    def b(self):
        return self._b


class _GsInfo(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("error", C.c_double), ("converged", C.c_int32), ("levels", C.c_uint64)]


class GaussSeidelResult:
    """What `gauss_seidel` of the heat example returns (heat.rs:103-139): `Ok((iterations, error))` when
    `converged`, `Err(error)` otherwise (then `iterations == max_iter`).  `levels` is the length of the
    longest chain of rows that have to be swept one after the other (device-side information)."""

    def __init__(self, info):
        self.converged = bool(info.converged)
        self.iterations = int(info.iterations)
        self.error = float(info.error)
        self.levels = int(info.levels)

    def __repr__(self):
        return ("Ok((%d, %r))" % (self.iterations, self.error)) if self.converged else "Err(%r)" % self.error


def gauss_seidel(mat, x, rhs, max_iter, eps, stream=None):
    """gauss_seidel(mat, x, rhs, max_iter, eps) (heat.rs:103-139): mat a square CSR DeviceCsMat, x (start vector,
    overwritten with the result like the reference's `mut x`) and rhs DeviceVecs.  Dimension mismatch and a row
    without a diagonal entry raise, like the reference's asserts / `diag.unwrap()`."""
    if x.n != rhs.n:
        raise _ffi.SprsHipError(_ffi.DIM_MISMATCH, "Dimension mismatch")
    info = _GsInfo()
    check(lib.sprs_hip_gauss_seidel_f64(mat._h, C.c_void_p(x.ptr), C.c_void_p(rhs.ptr), x.n, int(max_iter), float(eps),
                                        C.byref(info), C.c_void_p(int(stream) if stream else 0)))
    return GaussSeidelResult(info)
