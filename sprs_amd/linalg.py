"""Host mirror of sprs::linalg::bicgstab (sprs/src/sparse/linalg/bicgstab.rs) over the device path:
the solver loop runs behind `sprs_hip_bicgstab_f64` with every vector resident in HBM (two SpMVs,
four fused element-wise kernels and three dot launches per iteration; include/sprs_hip.h)."""
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
