"""sprs_amd — MI355X (gfx950) backend for the sprs CSR SpMV / SpGEMM hot path.

Product code: libsprs_hip.so (sprs_amd/csrc, hand-written HIP behind the C ABI
of include/sprs_hip.h) and this thin host mirror of the reference interface
(`prod`, `smmp`, DeviceCsMat/DeviceVec with `*`).  Importing fails loudly when
the shared library is absent; nothing here computes on the CPU.
"""
from . import _ffi
from ._ffi import CSC, CSR, SprsHipError
from .device import DeviceCsMat, DeviceVec
from . import prod, smmp


def device_count():
    import ctypes as C
    n = C.c_int32()
    st = _ffi.lib.sprs_hip_device_count(C.byref(n))
    return n.value if st == _ffi.OK else 0


def pool_trim():
    """Hand the pooled result blocks back to the driver (include/sprs_hip.h: sprs_hip_pool_trim);
    returns the number of bytes released."""
    import ctypes
    freed = ctypes.c_uint64(0)
    _ffi.check(_ffi.lib.sprs_hip_pool_trim(ctypes.byref(freed)))
    return int(freed.value)


def set_option(name, value):
    _ffi.check(_ffi.lib.sprs_hip_set_option(name.encode(), int(value)))


def get_option(name):
    import ctypes as C
    v = C.c_int64()
    _ffi.check(_ffi.lib.sprs_hip_get_option(name.encode(), C.byref(v)))
    return v.value


def version():
    return _ffi.lib.sprs_hip_version().decode()
