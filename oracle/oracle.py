"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes front end of the CPU oracle (oracle/liboracle.so, built from
sprs_oracle.c by oracle/Makefile): a C restatement of sprs' CPU path
  prod::mul_acc_mat_vec_csr   sprs/src/sparse/prod.rs:103-127
  smmp::symbolic / numeric    sprs/src/sparse/smmp.rs:81-131, 151-189
  smmp::mul_csr_csr           sprs/src/sparse/smmp.rs:196-416
plus CsMat::eye (csmat.rs:416-426), grid_laplacian (examples/heat.rs:45-80)
and raw::convert_mat_storage (csmat.rs:1782-1829).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (sprs_amd) never does.

Parity is pinned by tests/test_oracle_golden.py against the reference's own
golden vectors.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

OK, DIM_MISMATCH, STORAGE_MISMATCH, INDEX_OVERFLOW, BAD_STRUCTURE = 0, 1, 2, 3, 4


class OracleError(Exception):
    """Raised where the reference panics; `.code` is the oracle status."""

    _TEXT = {
        DIM_MISMATCH: "Dimension mismatch",
        STORAGE_MISMATCH: "Storage mismatch",
        INDEX_OVERFLOW: "Index type is not large enough to hold",
        BAD_STRUCTURE: "bad compressed structure",
    }

    def __init__(self, code):
        self.code = code
        super().__init__(self._TEXT.get(code, "oracle error %d" % code))


def build(force=False):
    """Compile liboracle.so (gcc) if missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("sprs_oracle.c", "sprs_oracle_impl.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_free.argtypes = [C.c_void_p]
        _lib.oracle_free.restype = None
    return _lib


def num_procs():
    return int(lib().oracle_num_procs())


def num_physical_cores():
    """what num_cpus::get_physical() counts on Linux: distinct (physical id, core id) pairs of /proc/cpuinfo
    (falls back to the logical count, as the crate does, when the file does not say)"""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        return len(cores) or num_procs()
    except OSError:
        return num_procs()


def automatic_physical_threads(a_nnz, b_nnz):
    """ThreadingStrategy::AutomaticPhysical (smmp.rs:26-31, 210-227): the Automatic rule min(cpus, (nnzA + nnzB) / 8128)
    with the number of PHYSICAL cores; returned as the Fixed(n) it resolves to"""
    return max(1, min(num_physical_cores(), (int(a_nnz) + int(b_nnz)) // 8128))


def _suffix(indices, indptr):
    ib, pb = indices.dtype.itemsize, indptr.dtype.itemsize
    suf = {(8, 8): "u64u64", (4, 4): "u32u32", (4, 8): "u32u64"}.get((ib, pb))
    if suf is None:
        raise TypeError("unsupported (index,indptr) widths %d/%d" % (ib, pb))
    return suf


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(code):
    if code != OK:
        raise OracleError(code)


def _canon(indptr, indices, data=None):
    indptr = np.ascontiguousarray(indptr)
    indices = np.ascontiguousarray(indices)
    if indptr.dtype.kind not in "iu" or indices.dtype.kind not in "iu":
        raise TypeError("integer index arrays expected")
    if data is not None:
        data = np.ascontiguousarray(data, dtype=np.float64)
    return indptr, indices, data


def mul_acc_mat_vec_csr(shape, indptr, indices, data, x, y, threads=1):
    """y += A x in place (prod.rs:103-127).  threads>1 selects the OpenMP
    row-split variant, which is NOT in the reference (SpMV there is serial)."""
    rows, cols = shape
    indptr, indices, data = _canon(indptr, indices, data)
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert y.dtype == np.float64 and y.flags.c_contiguous
    suf = _suffix(indices, indptr)
    u64 = C.c_uint64
    if threads == 1:
        f = getattr(lib(), "oracle_mul_acc_mat_vec_csr_" + suf)
        code = f(u64(rows), u64(cols), _p(indptr), _p(indices), _p(data), _p(x), u64(x.size),
                 _p(y), u64(y.size))
    else:
        f = getattr(lib(), "oracle_mul_acc_mat_vec_csr_omp_" + suf)
        code = f(u64(rows), u64(cols), _p(indptr), _p(indices), _p(data), _p(x), u64(x.size),
                 _p(y), u64(y.size), C.c_int(threads))
    _chk(code)
    return y


def _take(ptr, n, dtype):
    """Copy n items out of a malloc'd oracle buffer and free it."""
    n = int(n)
    if n:
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=dtype, count=n).copy()
    else:
        out = np.zeros(0, dtype=dtype)
    lib().oracle_free(ptr)
    return out


def symbolic(a_shape, a_indptr, a_indices, b_shape, b_indptr, b_indices):
    """smmp::symbolic on one chunk (smmp.rs:81-131) -> (c_indptr, c_indices)."""
    a_indptr, a_indices, _ = _canon(a_indptr, a_indices)
    b_indptr, b_indices, _ = _canon(b_indptr, b_indices)
    suf = _suffix(a_indices, a_indptr)
    assert suf == _suffix(b_indices, b_indptr)
    c_indptr = np.zeros(a_shape[0] + 1, dtype=a_indptr.dtype)
    out = C.c_void_p()
    nnz = C.c_uint64()
    u64 = C.c_uint64
    f = getattr(lib(), "oracle_symbolic_" + suf)
    _chk(f(u64(a_shape[0]), u64(a_shape[1]), _p(a_indptr), _p(a_indices), u64(b_shape[0]),
           u64(b_shape[1]), _p(b_indptr), _p(b_indices), _p(c_indptr), C.byref(out),
           C.byref(nnz)))
    return c_indptr, _take(out, nnz.value, a_indices.dtype)


def numeric(a_shape, a_indptr, a_indices, a_data, b_shape, b_indptr, b_indices, b_data,
            c_indptr, c_indices):
    """smmp::numeric on one chunk (smmp.rs:151-189) -> c_data."""
    a_indptr, a_indices, a_data = _canon(a_indptr, a_indices, a_data)
    b_indptr, b_indices, b_data = _canon(b_indptr, b_indices, b_data)
    c_indptr, c_indices, _ = _canon(c_indptr, c_indices)
    suf = _suffix(a_indices, a_indptr)
    c_data = np.zeros(c_indices.size, dtype=np.float64)
    u64 = C.c_uint64
    f = getattr(lib(), "oracle_numeric_" + suf)
    _chk(f(u64(a_shape[0]), u64(a_shape[1]), _p(a_indptr), _p(a_indices), _p(a_data),
           u64(b_shape[0]), u64(b_shape[1]), _p(b_indptr), _p(b_indices), _p(b_data),
           _p(c_indptr), _p(c_indices), _p(c_data)))
    return c_data


def mul_csr_csr(a_shape, a_indptr, a_indices, a_data, b_shape, b_indptr, b_indices, b_data,
                threads=0, return_threads=False):
    """smmp::mul_csr_csr (smmp.rs:196-416).  threads=0 is
    ThreadingStrategy::Automatic, threads=n is Fixed(n).
    Returns ((rows, cols), indptr, indices, data)."""
    a_indptr, a_indices, a_data = _canon(a_indptr, a_indices, a_data)
    b_indptr, b_indices, b_data = _canon(b_indptr, b_indices, b_data)
    suf = _suffix(a_indices, a_indptr)
    assert suf == _suffix(b_indices, b_indptr), "operands must share index types (smmp.rs:196-199)"
    ip, ix, dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz = C.c_uint64()
    used = C.c_int()
    u64 = C.c_uint64
    f = getattr(lib(), "oracle_mul_csr_csr_" + suf)
    _chk(f(u64(a_shape[0]), u64(a_shape[1]), _p(a_indptr), _p(a_indices), _p(a_data),
           u64(b_shape[0]), u64(b_shape[1]), _p(b_indptr), _p(b_indices), _p(b_data),
           C.c_int(threads), C.byref(ip), C.byref(ix), C.byref(dt), C.byref(nnz),
           C.byref(used)))
    c_indptr = _take(ip, a_shape[0] + 1, a_indptr.dtype)
    c_indices = _take(ix, nnz.value, a_indices.dtype)
    c_data = _take(dt, nnz.value, np.float64)
    res = ((a_shape[0], b_shape[1]), c_indptr, c_indices, c_data)
    return res + (used.value,) if return_threads else res


def eye(dim, idx_dtype=np.uint64, ptr_dtype=np.uint64):
    """CsMat::eye (csmat.rs:416-426)."""
    indptr = np.zeros(dim + 1, dtype=ptr_dtype)
    indices = np.zeros(dim, dtype=idx_dtype)
    data = np.zeros(dim, dtype=np.float64)
    getattr(lib(), "oracle_eye_" + _suffix(indices, indptr))(
        C.c_uint64(dim), _p(indptr), _p(indices), _p(data))
    return (dim, dim), indptr, indices, data


def grid_laplacian(rows, cols, idx_dtype=np.uint64, ptr_dtype=np.uint64):
    """grid_laplacian (examples/heat.rs:45-80)."""
    nv = rows * cols
    border = nv - max(rows - 2, 0) * max(cols - 2, 0)
    nnz = border + 5 * (nv - border)
    indptr = np.zeros(nv + 1, dtype=ptr_dtype)
    indices = np.zeros(nnz, dtype=idx_dtype)
    data = np.zeros(nnz, dtype=np.float64)
    f = getattr(lib(), "oracle_grid_laplacian_" + _suffix(indices, indptr))
    f.restype = C.c_uint64
    got = f(C.c_uint64(rows), C.c_uint64(cols), _p(indptr), _p(indices), _p(data))
    assert got == nnz
    return (nv, nv), indptr, indices, data


def convert_storage(outer, inner, indptr, indices, data, mat_rows=None):
    """raw::convert_mat_storage (csmat.rs:1782-1829): CSR(outer x inner) ->
    the CSR arrays of the transpose (== CSC arrays of the same matrix)."""
    indptr, indices, data = _canon(indptr, indices, data)
    o_indptr = np.zeros(inner + 1, dtype=indptr.dtype)
    o_indices = np.zeros(indices.size, dtype=indices.dtype)
    o_data = np.zeros(indices.size, dtype=np.float64)
    u64 = C.c_uint64
    f = getattr(lib(), "oracle_convert_storage_" + _suffix(indices, indptr))
    _chk(f(u64(outer), u64(inner), u64(outer if mat_rows is None else mat_rows), _p(indptr),
           _p(indices), _p(data), _p(o_indptr), _p(o_indices), _p(o_data)))
    return o_indptr, o_indices, o_data


def check_structure(inner, outer, indptr, indices):
    """utils::check_compressed_structure (sparse.rs:300-358)."""
    indptr, indices, _ = _canon(indptr, indices)
    if indptr.size != outer + 1:
        raise OracleError(BAD_STRUCTURE)
    f = getattr(lib(), "oracle_check_structure_" + _suffix(indices, indptr))
    _chk(f(C.c_uint64(inner), C.c_uint64(outer), _p(indptr), _p(indices),
           C.c_uint64(indices.size)))


def csmat_mul_csmat(lhs, rhs, threads=0):
    """csmat_mul_csmat storage dispatch (csmat.rs:1895-1949).  Operands are
    dicts {storage:'CSR'|'CSC', shape:(rows,cols), indptr, indices, data};
    result has the lhs' storage."""
    def t_view(m):   # transpose_view: free, flips the storage tag (csmat.rs:982-991)
        return dict(storage="CSC" if m["storage"] == "CSR" else "CSR",
                    shape=(m["shape"][1], m["shape"][0]),
                    indptr=m["indptr"], indices=m["indices"], data=m["data"])

    def other(m):    # to_other_storage (csmat.rs:1405-1426)
        rows, cols = m["shape"]
        outer, inner = (rows, cols) if m["storage"] == "CSR" else (cols, rows)
        ip, ix, dt = convert_storage(outer, inner, m["indptr"], m["indices"], m["data"],
                                     mat_rows=rows)
        return dict(storage="CSC" if m["storage"] == "CSR" else "CSR", shape=m["shape"],
                    indptr=ip, indices=ix, data=dt)

    def csr_csr(a, b):
        shape, ip, ix, dt = mul_csr_csr(a["shape"], a["indptr"], a["indices"], a["data"],
                                        b["shape"], b["indptr"], b["indices"], b["data"],
                                        threads=threads)
        return dict(storage="CSR", shape=shape, indptr=ip, indices=ix, data=dt)

    ls, rs = lhs["storage"], rhs["storage"]
    if (ls, rs) == ("CSR", "CSR"):
        return csr_csr(lhs, rhs)
    if (ls, rs) == ("CSR", "CSC"):
        return csr_csr(lhs, other(rhs))
    if (ls, rs) == ("CSC", "CSR"):
        return t_view(csr_csr(t_view(other(rhs)), t_view(lhs)))
    return t_view(csr_csr(t_view(rhs), t_view(lhs)))


class BicgstabInfo(C.Structure):
    _fields_ = [("iteration_count", C.c_uint64), ("soft_restart_count", C.c_uint64),
                ("hard_restart_count", C.c_uint64), ("err", C.c_double), ("rho", C.c_double),
                ("converged", C.c_int32)]


def bicgstab(shape, indptr, indices, data, x0, b, tol, max_iter, soft_restart_threshold=0.1, storage="CSR"):
    """BiCGSTAB::solve (sprs/src/sparse/linalg/bicgstab.rs:148-171) on dense vectors; a CSC operand is
    converted first (its product adds ascending k as well).  Returns (x, info dict)."""
    n = shape[0]
    assert shape[0] == shape[1] == np.asarray(x0).size == np.asarray(b).size
    indptr, indices, data = _canon(indptr, indices, data)
    if storage == "CSC":
        indptr, indices, data = convert_storage(n, n, indptr, indices, data)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    info = BicgstabInfo()
    f = getattr(lib(), "oracle_bicgstab_" + _suffix(indices, indptr))
    _chk(f(C.c_uint64(n), _p(indptr), _p(indices), _p(data), _p(x0), _p(b), C.c_double(tol), C.c_uint64(max_iter),
           C.c_double(soft_restart_threshold), _p(x), C.byref(info)))
    return x, {k: getattr(info, k) for k, _ in BicgstabInfo._fields_}


class GaussSeidelInfo(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("error", C.c_double), ("converged", C.c_int32)]


def gauss_seidel(shape, indptr, indices, data, x, rhs, max_iter, eps):
    """gauss_seidel (sprs/examples/heat.rs:103-139) on a CSR matrix: sweeps in place over a COPY of x.
    Returns (x, info dict): converged 1 = Ok((iterations, error)), 0 = Err(error)."""
    n = shape[0]
    assert shape[0] == shape[1] == np.asarray(x).size == np.asarray(rhs).size
    indptr, indices, data = _canon(indptr, indices, data)
    x = np.array(x, dtype=np.float64, copy=True)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    info = GaussSeidelInfo()
    f = getattr(lib(), "oracle_gauss_seidel_" + _suffix(indices, indptr))
    _chk(f(C.c_uint64(n), _p(indptr), _p(indices), _p(data), _p(x), _p(rhs), C.c_uint64(max_iter), C.c_double(eps),
           C.byref(info)))
    return x, {k: getattr(info, k) for k, _ in GaussSeidelInfo._fields_}


def triplets_to_cs(shape, row_inds, col_inds, data, storage="CSR", idx_dtype=np.uint64):
    """TriMatIter::into_cs (sprs/src/sparse/triplet_iter.rs:127-224): sort the triplets by (outer, inner),
    fold equal (row, col) neighbours with `slot = slot + next` (left to right, :168-171), fill indptr.
    The reference's sort is UNSTABLE, so the order inside a group of duplicates — hence the rounding of a sum
    of three or more — is unspecified there; this restatement takes the stable order (triplet order), which
    is what the device path produces.  Explicit zeros and sums that cancel stay stored.
    Returns (indptr u64, indices idx_dtype, data)."""
    rows, cols = shape
    r = np.asarray(row_inds, dtype=np.int64)
    c = np.asarray(col_inds, dtype=np.int64)
    v = np.asarray(data, dtype=np.float64)
    outer, inner, n_outer = (r, c, rows) if storage == "CSR" else (c, r, cols)
    order = np.lexsort((inner, outer))                       # stable: ties keep triplet order
    outer, inner, v = outer[order], inner[order], v[order]
    out_outer, out_inner, out_v = [], [], []
    for k in range(v.size):                                   # the reference's loop, slot by slot
        if k and outer[k] == outer[k - 1] and inner[k] == inner[k - 1]:
            out_v[-1] = out_v[-1] + v[k]
        else:
            out_outer.append(outer[k])
            out_inner.append(inner[k])
            out_v.append(v[k])
    indptr = np.zeros(n_outer + 1, dtype=np.uint64)
    np.add.at(indptr, np.asarray(out_outer, dtype=np.int64) + 1, 1)
    indptr = np.cumsum(indptr).astype(np.uint64)
    return indptr, np.asarray(out_inner, dtype=idx_dtype), np.asarray(out_v, dtype=np.float64)

