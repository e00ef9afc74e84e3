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
