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
