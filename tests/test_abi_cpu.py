"""CPU-only checks of the drop-in boundary: libsprs_hip.so loads, exports every
symbol include/sprs_hip.h declares, reports errors the documented way, and
never computes without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sprs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sprs_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from sprs_amd import _ffi
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_ffi.lib, n), "libsprs_hip.so does not export %s" % n
        assert n in _ffi.SIGNATURES, "sprs_amd/_ffi.py has no signature for %s" % n
    assert set(_ffi.SIGNATURES) <= set(names), "binding declares symbols the header does not"


def test_rust_sys_crate_declares_the_same_symbols():
    """rust/sprs-hip-sys (source only: no Rust toolchain in this image) must not drift from the header"""
    rs = open(os.path.join(ROOT, "rust", "sprs-hip-sys", "src", "lib.rs")).read()
    assert sorted(set(re.findall(r"pub fn (sprs_hip_[a-z0-9_]+)", rs))) == declared_symbols()
    wrapper = open(os.path.join(ROOT, "rust", "sprs-hip", "src", "lib.rs")).read()
    for used in set(re.findall(r"sys::(sprs_hip_[a-z0-9_]+)", wrapper)):
        assert used in rs, used


def test_status_codes_match_header():
    from sprs_amd import _ffi
    src = open(os.path.join(ROOT, "include", "sprs_hip.h")).read()
    codes = dict(re.findall(r"#define\s+SPRS_HIP_([A-Z_]+)\s+(\d+)", src))
    for name, val in codes.items():
        if name in ("CSR", "CSC", "H"):
            continue
        assert getattr(_ffi, name) == int(val), name
    assert _ffi.CSR == int(codes["CSR"]) and _ffi.CSC == int(codes["CSC"])


def test_version_and_options():
    import sprs_amd
    assert "gfx950" in sprs_amd.version()
    assert sprs_amd.get_option("spmv_xcs") == 0 and sprs_amd.get_option("spmv_xcs_split") >= 2
    with pytest.raises(sprs_amd.SprsHipError) as e:
        sprs_amd.set_option("no_such_option", 1)
    assert e.value.status == sprs_amd._ffi.INVALID_ARG and "no_such_option" in str(e.value)
    with pytest.raises(sprs_amd.SprsHipError):
        sprs_amd.set_option("spmv_xcs", 7)


def test_release_library_rejects_developer_switches():
    """spgemm_debug / spmv_xmask (timing experiments with WRONG results) and spgemm_prof exist only in builds with
    -DSPRS_HIP_DEVTOOLS (make DEVTOOLS=1); the library the product ships is built without and must refuse them."""
    import sprs_amd
    assert sprs_amd.get_option("devtools") == 0, "the in-tree libsprs_hip.so must be the release build"
    for name, val in (("spgemm_debug", 1), ("spmv_xmask", 1023), ("spgemm_prof", 1)):
        with pytest.raises(sprs_amd.SprsHipError) as e:
            sprs_amd.set_option(name, val)
        assert e.value.status == sprs_amd._ffi.INVALID_ARG and "developer switch" in str(e.value)


def test_argument_checks_need_no_device():
    from sprs_amd import _ffi
    h = C.c_void_p()
    ip = np.array([0, 1], dtype=np.uint64)
    ix = np.array([0], dtype=np.uint64)
    dt = np.ones(1)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    st = _ffi.lib.sprs_hip_csmat_upload(C.byref(h), 0, 1, 1, vp(ip), 3, vp(ix), 8, vp(dt), 1)
    assert st == _ffi.INVALID_ARG and b"4 or 8" in _ffi.lib.sprs_hip_last_error()
    st = _ffi.lib.sprs_hip_csmat_upload(C.byref(h), 7, 1, 1, vp(ip), 8, vp(ix), 8, vp(dt), 1)
    assert st == _ffi.INVALID_ARG
    # structure validation happens on the host, before any device work (sparse.rs:300-358)
    bad = np.array([0, 2], dtype=np.uint64)
    ix2 = np.array([1, 0], dtype=np.uint64)
    st = _ffi.lib.sprs_hip_csmat_upload(C.byref(h), 0, 1, 2, vp(bad), 8, vp(ix2), 8, vp(np.ones(2)), 1)
    assert st == _ffi.BAD_STRUCTURE and b"not sorted" in _ffi.lib.sprs_hip_last_error()
    # 2-byte index types (u16 / i16, indexing.rs:124-130): a NULL indices / data pointer with entries to read is refused
    # before the widening loop touches it (it used to crash)
    ip16 = np.array([0, 1], dtype=np.uint16)
    ix16 = np.array([0], dtype=np.uint16)
    st = _ffi.lib.sprs_hip_csmat_upload(C.byref(h), 0, 1, 1, vp(ip16), 2, None, 2, vp(dt), 1)
    assert st == _ffi.INVALID_ARG and b"NULL indices" in _ffi.lib.sprs_hip_last_error()
    st = _ffi.lib.sprs_hip_csmat_upload(C.byref(h), 0, 1, 1, vp(ip16), 2, vp(ix16), 2, None, 1)
    assert st == _ffi.INVALID_ARG
    st = _ffi.lib.sprs_hip_spmv_f64(None, None, 0, None, 0, 0, None)
    assert st == _ffi.INVALID_ARG
    st = _ffi.lib.sprs_hip_spmv_f64_host(2, 2, vp(ip), 8, vp(ix), 8, vp(dt), vp(dt), 1, vp(dt), 2, 0)
    assert st == _ffi.DIM_MISMATCH and _ffi.lib.sprs_hip_last_error() == b"Dimension mismatch"


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute."""
    import sprs_amd
    if sprs_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    from sprs_amd.device import DeviceCsMat
    with pytest.raises(sprs_amd.SprsHipError) as e:
        DeviceCsMat.eye(4)
    assert e.value.status == sprs_amd._ffi.NO_DEVICE


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under sprs_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sprs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(dirpath, f)
