"""ctypes binding of libsprs_hip.so (include/sprs_hip.h).

This is the Python stand-in for the `sprs-hip-sys` crate: raw declarations
only.  There is NO CPU fallback: if the shared library is missing the import
fails loudly, and if no gfx950 device is usable every compute entry point
returns SPRS_HIP_NO_DEVICE, surfaced as SprsHipError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPRS_HIP_LIBRARY: explicit path of the shared library to bind (deployments that keep it elsewhere; the
# kernel-logic tests point it at the emulator build of the same sources, tests/emu).  No fallback either way.
LIB_PATH = os.environ.get("SPRS_HIP_LIBRARY") or os.path.join(_HERE, "libsprs_hip.so")

OK, DIM_MISMATCH, STORAGE_MISMATCH, INDEX_OVERFLOW, BAD_STRUCTURE, INVALID_ARG, \
    OUT_OF_MEMORY, HIP_ERROR, NO_DEVICE = range(9)
CSR, CSC = 0, 1
ROW_MAJOR, COL_MAJOR = 0, 1
ROUTE_RCCL, ROUTE_PEER = 0, 1

STATUS_NAMES = {
    OK: "OK", DIM_MISMATCH: "DIM_MISMATCH", STORAGE_MISMATCH: "STORAGE_MISMATCH",
    INDEX_OVERFLOW: "INDEX_OVERFLOW", BAD_STRUCTURE: "BAD_STRUCTURE", INVALID_ARG: "INVALID_ARG",
    OUT_OF_MEMORY: "OUT_OF_MEMORY", HIP_ERROR: "HIP_ERROR", NO_DEVICE: "NO_DEVICE",
}

u64, i32, i64, vp = C.c_uint64, C.c_int32, C.c_int64, C.c_void_p
P = C.POINTER

# name -> (restype, argtypes); must list every symbol include/sprs_hip.h declares
SIGNATURES = {
    "sprs_hip_last_error": (C.c_char_p, []),
    "sprs_hip_last_hip_code": (i32, []),
    "sprs_hip_version": (C.c_char_p, []),
    "sprs_hip_device_count": (i32, [P(i32)]),
    "sprs_hip_set_device": (i32, [i32]),
    "sprs_hip_malloc": (i32, [P(vp), u64]),
    "sprs_hip_free": (i32, [vp]),
    "sprs_hip_memcpy_h2d": (i32, [vp, vp, u64]),
    "sprs_hip_memcpy_d2h": (i32, [vp, vp, u64]),
    "sprs_hip_memcpy_d2d": (i32, [vp, vp, u64, vp]),
    "sprs_hip_memset": (i32, [vp, i32, u64, vp]),
    "sprs_hip_synchronize": (i32, [vp]),
    "sprs_hip_pool_trim": (i32, [P(u64)]),
    "sprs_hip_mul_acc_mat_vec_csc_f64": (i32, [vp, vp, u64, vp, u64, vp]),
    "sprs_hip_csmat_mul_vec_f64": (i32, [vp, vp, u64, vp, u64, vp]),
    "sprs_hip_csmat_mulacc_dense_f64": (i32, [vp, vp, u64, u64, i32, u64, vp, u64, i32, u64, i32, vp]),
    "sprs_hip_csmat_mul_dense_f64": (i32, [vp, vp, u64, u64, i32, u64, vp, P(i32), vp]),
    "sprs_hip_dense_dot_csmat_f64": (i32, [vp, u64, u64, i32, u64, vp, vp, P(i32), vp]),
    "sprs_hip_dist_comm_count": (i32, [vp, P(i32)]),
    "sprs_hip_spgemm_symbolic": (i32, [vp, vp, P(vp)]),
    "sprs_hip_spgemm_numeric": (i32, [vp, vp, vp]),
    "sprs_hip_spgemm_plan_create": (i32, [vp, vp, P(vp)]),
    "sprs_hip_spgemm_plan_nnz": (i32, [vp, P(u64)]),
    "sprs_hip_spgemm_plan_structure": (i32, [vp, vp, vp, P(vp)]),
    "sprs_hip_spgemm_plan_product": (i32, [vp, vp, vp, P(vp)]),
    "sprs_hip_spgemm_plan_numeric": (i32, [vp, vp, vp, vp]),
    "sprs_hip_spgemm_plan_free": (i32, [vp]),
    "sprs_hip_bicgstab_f64": (i32, [vp, vp, vp, u64, C.c_double, u64, C.c_double, vp, vp, vp]),
    "sprs_hip_gauss_seidel_f64": (i32, [vp, vp, vp, u64, u64, C.c_double, vp, vp]),
    "sprs_hip_csmat_upload": (i32, [P(vp), i32, u64, u64, vp, i32, vp, i32, vp, i32]),
    "sprs_hip_csmat_wrap_device": (i32, [P(vp), i32, u64, u64, u64, vp, i32, vp, i32, vp]),
    "sprs_hip_csmat_info": (i32, [vp, P(u64), P(u64), P(u64), P(i32), P(i32), P(i32)]),
    "sprs_hip_csmat_device_ptrs": (i32, [vp, P(vp), P(vp), P(vp)]),
    "sprs_hip_csmat_download": (i32, [vp, vp, vp, vp]),
    "sprs_hip_csmat_download_outer": (i32, [vp, u64, u64, vp, vp, vp, P(u64)]),
    "sprs_hip_csmat_slice_outer": (i32, [vp, u64, u64, P(vp)]),
    "sprs_hip_csmat_refresh": (i32, [vp]),
    "sprs_hip_csmat_prepare": (i32, [vp, vp]),
    "sprs_hip_csmat_spmv_plan_info": (i32, [vp, P(i32), P(u64)]),
    "sprs_hip_csmat_transpose_view": (i32, [vp, P(vp)]),
    "sprs_hip_csmat_free": (i32, [vp]),
    "sprs_hip_spmv_f64": (i32, [vp, vp, u64, vp, u64, i32, vp]),
    "sprs_hip_spmv_f64_host": (i32, [u64, u64, vp, i32, vp, i32, vp, vp, u64, vp, u64, i32]),
    "sprs_hip_spmm_rowmaj_f64": (i32, [vp, vp, u64, u64, u64, vp, u64, u64, i32, vp]),
    "sprs_hip_spgemm_f64": (i32, [vp, vp, P(vp)]),
    "sprs_hip_csmat_to_other_storage": (i32, [vp, P(vp)]),
    "sprs_hip_dist_unique_id": (i32, [vp]),
    "sprs_hip_dist_create": (i32, [P(vp), vp, i32, i32, u64, u64, P(u64), vp, i32]),
    "sprs_hip_dist_spmv_f64": (i32, [vp, vp, u64, vp, u64, vp]),
    "sprs_hip_dist_peer_handle": (i32, [vp, vp]),
    "sprs_hip_dist_peer_connect": (i32, [vp, vp, i32]),
    "sprs_hip_dist_set_route": (i32, [vp, i32]),
    "sprs_hip_dist_route": (i32, [vp, P(i32)]),
    "sprs_hip_dist_free": (i32, [vp]),
    "sprs_hip_csmat_mul_csmat": (i32, [vp, vp, P(vp)]),
    "sprs_hip_triplets_to_cs": (i32, [u64, u64, u64, vp, vp, i32, vp, i32, i32, i32, P(vp)]),
    "sprs_hip_set_option": (i32, [C.c_char_p, i64]),
    "sprs_hip_get_option": (i32, [C.c_char_p, P(i64)]),
}


class SprsHipError(RuntimeError):
    """A non-OK status from libsprs_hip.so.  The message of the contract
    violations is the reference's own panic text ("Dimension mismatch",
    "Storage mismatch", "Index type is not large enough to hold ...")."""

    def __init__(self, status, message, hip_code=0):
        self.status = status
        self.hip_code = hip_code
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "sprs_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C sprs_amd/csrc`.  There is no CPU fallback." % LIB_PATH)



def _preload_shared_hip_runtime():
    """One process must use ONE libamdhip64: PyTorch wheels bundle their own
    copy (same SONAME as /opt/rocm's).  If torch is installed, load its copy
    first so that libsprs_hip.so and torch share a single HIP runtime whatever
    the import order (bench.py and the multi-GPU path hand torch-owned device
    buffers to this library).  torch itself is NOT imported here."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:   # fall back to the system ROCm runtime (RUNPATH of the .so)
        pass


_preload_shared_hip_runtime()
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)   # AttributeError here == the .so does not match the header
    _f.restype = _res
    _f.argtypes = _args


def check(status):
    if status != OK:
        raise SprsHipError(status, lib.sprs_hip_last_error().decode("utf-8", "replace"),
                           lib.sprs_hip_last_hip_code())
