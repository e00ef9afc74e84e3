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


# ---- type-level drift guard: the Rust crates are never compiled here, so the only "compiler" their
# extern block ever sees is this comparison with the C header, argument by argument -------------------------------------------
_C_SCALARS = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "float": "f32",
              "char": "c_char", "void": "c_void", "int": "i32", "size_t": "usize", "uint8_t": "u8", "uint16_t": "u16"}


def _canon_c(decl):
    """'const sprs_hip_csmat *a' / 'uint64_t n' / 'sprs_hip_csmat **out' -> the Rust spelling of the type (name dropped)"""
    decl = decl.strip()
    stars = decl.count("*")
    words = [w for w in re.sub(r"[*]", " ", decl).split() if w not in ("struct",)]
    const = "const" in words
    words = [w for w in words if w != "const"]
    base = words[0]
    base = _C_SCALARS.get(base, base)                                 # opaque handles and info structs keep their names
    t = base
    for level in range(stars):
        # `const T *`: pointer to const; deeper levels (T **out) are *mut *mut in the crate: only the innermost level of a
        # const-qualified base is const
        t = ("*const " if (const and level == 0) else "*mut ") + t
    return t


def c_prototypes():
    src = open(os.path.join(ROOT, "include", "sprs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?[ *]+)\b(sprs_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        args = args.strip()
        alist = [] if args in ("", "void") else [_canon_c(a) for a in args.split(",")]
        protos[name] = (_canon_c(ret + " r") if ret.strip() != "void" else "()", alist)
    return protos


def rust_prototypes():
    rs = open(os.path.join(ROOT, "rust", "sprs-hip-sys", "src", "lib.rs")).read()
    rs = re.sub(r"//[^\n]*", "", rs)
    protos = {}
    for name, args, ret in re.findall(r"pub fn (sprs_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", rs):
        alist = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            alist.append(re.sub(r"\s+", " ", a.split(":", 1)[1].strip()))
        protos[name] = (re.sub(r"\s+", " ", ret.strip()) if ret else "()", alist)
    return protos


def test_rust_sys_crate_matches_the_header_type_by_type():
    c, r = c_prototypes(), rust_prototypes()
    assert sorted(c) == declared_symbols() and sorted(r) == sorted(c)
    for name in sorted(c):
        assert len(c[name][1]) == len(r[name][1]), "%s: %d arguments in the header, %d in rust/sprs-hip-sys" % (name, len(c[name][1]), len(r[name][1]))
        assert c[name][0] == r[name][0], "%s: returns %s in the header, %s in the crate" % (name, c[name][0], r[name][0])
        for i, (ct, rt) in enumerate(zip(c[name][1], r[name][1])):
            assert ct == rt, "%s, argument %d: header says %s, crate says %s" % (name, i, ct, rt)


def test_ctypes_binding_matches_the_header_type_by_type():
    """the same for sprs_amd/_ffi.py: arity, integer / float widths, pointer-ness (ctypes has one void pointer for every handle)"""
    from sprs_amd import _ffi
    c = c_prototypes()

    def klass(t):
        if t.startswith("*"):
            return "ptr"
        return {"c_char": "i8"}.get(t, t)

    def klass_ct(t):
        if t is None:
            return "()"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or (hasattr(t, "_type_") and not isinstance(t._type_, str)):
            return "ptr"
        return {C.c_int32: "i32", C.c_uint32: "u32", C.c_int64: "i64", C.c_uint64: "u64", C.c_double: "f64"}[t]

    for name, (res, args) in _ffi.SIGNATURES.items():
        assert len(args) == len(c[name][1]), "%s: %d arguments in the header, %d in _ffi.py" % (name, len(c[name][1]), len(args))
        assert klass(c[name][0]) == klass_ct(res), name
        for i, (ct, at) in enumerate(zip(c[name][1], args)):
            assert klass(ct) == klass_ct(at), "%s, argument %d: header says %s, _ffi.py says %s" % (name, i, ct, at)


def test_type_guard_catches_a_wrong_width():
    """the guard itself: a u32 where the header says uint64_t, a *mut where it says const, a missing argument"""
    assert _canon_c("const double *x_dev") == "*const f64" and _canon_c("sprs_hip_csmat **out") == "*mut *mut sprs_hip_csmat"
    assert _canon_c("uint64_t n") == "u64" and _canon_c("void *stream") == "*mut c_void" and _canon_c("const void *p") == "*const c_void"
    assert _canon_c("const sprs_hip_csmat *a") == "*const sprs_hip_csmat" and _canon_c("const char *name") == "*const c_char"


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
