"""Twin of `sprs::smmp` for device operands (sprs/src/sparse/smmp.rs)."""
import ctypes as C

from ._ffi import check, lib
from .device import DeviceCsMat


def mul_csr_csr(lhs, rhs):
    """smmp::mul_csr_csr (smmp.rs:196-416): C = lhs * rhs, all CSR, rows
    sorted, structural zeros kept.  Returns a new DeviceCsMat."""
    h = C.c_void_p()
    check(lib.sprs_hip_spgemm_f64(lhs._h, rhs._h, C.byref(h)))
    return DeviceCsMat(h.value)



def symbolic(a, b):
    """smmp::symbolic (smmp.rs:81-131): the structure of a * b — indptr and sorted indices, structural zeros
    kept — as a DeviceCsMat whose values are 0.0."""
    h = C.c_void_p()
    check(lib.sprs_hip_spgemm_symbolic(a._h, b._h, C.byref(h)))
    return DeviceCsMat(h.value)


def numeric(a, b, c):
    """smmp::numeric (smmp.rs:151-189): the values of a * b into c, which must have the product's structure
    (as `symbolic` returns it); BAD_STRUCTURE otherwise."""
    check(lib.sprs_hip_spgemm_numeric(a._h, b._h, c._h))
    return c


class SpgemmPlan:
    """The symbolic phase of a * b, kept (sprs_hip_spgemm_plan_*): what a caller of the reference keeps between
    smmp::symbolic and smmp::numeric (smmp.rs:81-131, 151-189) — here the cut of the work too, so that `numeric`
    launches the value kernels only.  a and b must stay alive and keep their structure; their values may change."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        h = C.c_void_p()
        check(lib.sprs_hip_spgemm_plan_create(a._h, b._h, C.byref(h)))
        self._h = h

    def nnz(self):
        n = C.c_uint64()
        check(lib.sprs_hip_spgemm_plan_nnz(self._h, C.byref(n)))
        return n.value

    def structure(self):
        """C with indptr and sorted indices, values 0.0 (smmp::symbolic's result)"""
        h = C.c_void_p()
        check(lib.sprs_hip_spgemm_plan_structure(self._h, self.a._h, self.b._h, C.byref(h)))
        return DeviceCsMat(h.value)

    def product(self):
        """C = a * b complete"""
        h = C.c_void_p()
        check(lib.sprs_hip_spgemm_plan_product(self._h, self.a._h, self.b._h, C.byref(h)))
        return DeviceCsMat(h.value)

    def numeric(self, c):
        """the values of a * b into c (structure as returned by structure() / product()); value kernels only"""
        check(lib.sprs_hip_spgemm_plan_numeric(self._h, self.a._h, self.b._h, c._h))
        return c

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:       # (lib is None while the interpreter shuts down)
            lib.sprs_hip_spgemm_plan_free(h)
            self._h = C.c_void_p(0)
