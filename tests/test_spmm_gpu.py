"""GPU parity tests of the SpMM path (prod::csr_mulacc_dense_rowmaj twin, prod.rs:189-214) against
the reference's golden vectors and the oracle (SpMM == SpMV per rhs column, prod.rs:274-298)."""
import numpy as np
import pytest

from conftest import IDX_COMBOS, as_csr
from helpers import ragged_csr, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def hip():
    import sprs_amd
    if sprs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (no CPU fallback exists)")
    return sprs_amd


def oracle_spmm(shape, ip, ix, dt, rhs, out0=None):
    from oracle import oracle
    out = np.zeros((shape[0], rhs.shape[1])) if out0 is None else out0.copy()
    for j in range(rhs.shape[1]):
        col = np.ascontiguousarray(out[:, j])
        oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, np.ascontiguousarray(rhs[:, j]), col)
        out[:, j] = col
    return out


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_golden_mul_csr_dense_rowmaj(hip, golden, idx, ptr):
    # prod.rs:502-542: eye(3) * eye(3); mat1 * mat_dense1 (exact); mat5 * mat_dense2 (eps 1e-8)
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat
    e = DeviceCsMat.eye(3)
    a = prod.DeviceMat.from_host(np.eye(3))
    res = prod.DeviceMat(3, 3)
    prod.csr_mulacc_dense_rowmaj(e, a, res)
    assert np.array_equal(res.to_host(), np.eye(3))
    for mk, dk, ek in (("mat1", "mat_dense1", "mat1_times_mat_dense1"), ("mat5", "mat_dense2", "mat5_times_mat_dense2")):
        m = DeviceCsMat.from_host(*as_csr(golden[mk], idx, ptr))
        b = prod.DeviceMat.from_host(np.array(golden[dk]))
        exp = np.array(golden[ek]["rows"])
        res = prod.DeviceMat(exp.shape[0], exp.shape[1])
        prod.csr_mulacc_dense_rowmaj(m, b, res)
        assert np.all(np.abs(res.to_host() - exp) <= golden[ek]["epsilon"])
        c = (m * b).to_host()                                  # `&a * &b` (csmat.rs:1989-2048)
        assert np.all(np.abs(c - exp) <= golden[ek]["epsilon"])
        prod.csr_mulacc_dense_rowmaj(m, b, res)                # accumulates
        assert np.all(np.abs(res.to_host() - 2 * exp) <= 2 * golden[ek]["epsilon"] + 1e-12)


def test_golden_mul_csc_dense(hip, golden):
    """prod.rs:545-597: mul_csc_dense_rowmaj, mul_csc_dense_colmaj, mul_csr_dense_colmaj — mat1 (CSC / CSR) * mat_dense1 in both
    layouts, the explicit kernels and `&a * &b`; and the (CSC, cols < 8) / (CSC, cols >= 8) arms of the dispatch on wider data"""
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat, DeviceVec, CSC
    exp = np.array(golden["mat1_times_mat_dense1"]["rows"])
    dense = np.array(golden["mat_dense1"])
    g = golden["mat1_csc"]
    u = lambda v: np.array(v, dtype=np.uint64)
    a_csc = DeviceCsMat.from_host(tuple(g["shape"]), u(g["indptr"]), u(g["indices"]), np.array(g["data"]), storage=CSC)
    a_csr = DeviceCsMat.from_host(*as_csr(golden["mat1"], np.uint64, np.uint64))
    b = prod.DeviceMat.from_host(dense)
    res = prod.DeviceMat(5, 5)
    prod.csc_mulacc_dense_rowmaj(a_csc, b, res)                           # mul_csc_dense_rowmaj
    assert np.array_equal(res.to_host(), exp)
    assert np.array_equal((a_csc * b).to_host(), exp)
    for a, kernel in ((a_csc, prod.csc_mulacc_dense_colmaj), (a_csr, prod.csr_mulacc_dense_colmaj)):   # the two colmaj tests
        cols_in = [DeviceVec.from_host(np.ascontiguousarray(dense[:, j])) for j in range(5)]
        cols_out = [DeviceVec.zeros(5) for _ in range(5)]
        kernel(a, cols_in, cols_out)
        assert np.array_equal(np.stack([c.to_host() for c in cols_out], axis=1), exp)
    with pytest.raises(hip.SprsHipError) as e:                            # assert!(lhs.is_csc(), "Storage mismatch")
        prod.csc_mulacc_dense_rowmaj(a_csr, b, res)
    assert e.value.status == hip._ffi.STORAGE_MISMATCH
    with pytest.raises(hip.SprsHipError) as e:
        prod.csc_mulacc_dense_rowmaj(a_csc, prod.DeviceMat(4, 5), res)
    assert e.value.status == hip._ffi.DIM_MISMATCH
    # wider: a random CSC matrix times 3 and 11 columns against the oracle on its CSR form
    import scipy.sparse as sp
    rng = np.random.default_rng(8)
    m = sp.random(300, 200, density=0.05, random_state=4, format="csc")
    m.sort_indices()
    d = DeviceCsMat.from_host((300, 200), u(m.indptr), u(m.indices), m.data, storage=CSC)
    mr = m.tocsr()
    mr.sort_indices()
    for k in (3, 11):
        rhs = rng.standard_normal((200, k))
        got = (d * prod.DeviceMat.from_host(rhs)).to_host()
        ref = oracle_spmm((300, 200), u(mr.indptr), u(mr.indices), mr.data, rhs)
        assert rel_err(got, ref) <= 1e-10


@pytest.fixture(params=[(0, 1, 2), (0, 1, 1), (0, 0, 2), (2048, 1, 2)], ids=["stream", "stream-relaid-rhs", "chunks", "entry-order"])
def long_row(request, hip):
    """the summation modes of spmm.hip: tiles of 256 consecutive entries per wave (default) — gathering from the rhs itself or
    from its re-laid-out copy (option spmm_relayout) —, every row by 512-entry chunks (option spmm_stream = 0: the kernel of
    rounds 1-3), or rows of <= L entries in the reference's own entry order (option spmm_long_row = L: bit-identical to
    prod.rs:203-210)"""
    hip.set_option("spmm_long_row", request.param[0])
    hip.set_option("spmm_stream", request.param[1])
    hip.set_option("spmm_relayout", request.param[2])
    yield request.param[0]
    hip.set_option("spmm_long_row", -1)
    hip.set_option("spmm_stream", 1)
    hip.set_option("spmm_relayout", 0)


@pytest.mark.parametrize("k", [1, 3, 8, 16, 33, 64, 100])
def test_rmat_vs_oracle(hip, k, long_row):
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat
    n = 30000
    indptr, indices, data = gen.rmat_csr(n, 10, seed=13)        # rows longer than one 512-entry chunk too
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    assert int(np.diff(ip.astype(np.int64)).max()) > 1500
    rng = np.random.default_rng(k)
    rhs = rng.random((n, k)) + 0.5
    a = DeviceCsMat.from_host((n, n), ip, ix, dt)
    got = (a * prod.DeviceMat.from_host(rhs)).to_host()
    ref = oracle_spmm((n, n), ip, ix, dt, rhs)
    assert rel_err(got, ref) <= TOL
    lens = np.diff(ip.astype(np.int64))
    short = (lens <= long_row) | (lens <= 1)                    # rows of <= L entries are summed in entry order, the
    assert np.array_equal(got[short], ref[short])               # reference's own (prod.rs:203-210): bit for bit
    empty = np.diff(ip.astype(np.int64)) == 0
    assert empty.any() and np.all(got[empty] == 0.0)
    out0 = rng.random((n, k))
    res = prod.DeviceMat.from_host(out0)
    prod.csr_mulacc_dense_rowmaj(a, prod.DeviceMat.from_host(rhs), res)
    got2 = res.to_host()
    ref2 = oracle_spmm((n, n), ip, ix, dt, rhs, out0)
    assert rel_err(got2, ref2) <= TOL
    assert np.array_equal(got2[short], ref2[short])             # the accumulate form starts from out[i, j], as there
    assert np.array_equal(got2[empty], out0[empty])             # empty rows untouched
    again = (a * prod.DeviceMat.from_host(rhs)).to_host()
    assert np.array_equal(got, again)                           # deterministic


@pytest.mark.parametrize("k", [2, 16, 40])
def test_long_rows_take_the_chunk_kernels(hip, k, long_row):
    """rows above option spmm_long_row are cut into 512-entry chunks (or, in the stream mode, cross several 256-entry tiles):
    same values up to the summation order"""
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat
    rng = np.random.default_rng(100 + k)
    n, m = 300, 9000
    lens = rng.integers(0, 40, n)
    lens[[3, 150, 299]] = [2049, 7000, 4096]
    lens[[4, 151]] = [2048, 0]
    ip = np.zeros(n + 1, np.uint64)
    ip[1:] = np.cumsum(lens)
    ix = np.concatenate([np.sort(rng.choice(m, int(l), replace=False)) for l in lens]).astype(np.uint64)
    dt = rng.standard_normal(ix.size)
    rhs = rng.standard_normal((m, k))
    a = DeviceCsMat.from_host((n, m), ip, ix, dt)
    got = (a * prod.DeviceMat.from_host(rhs)).to_host()
    ref = oracle_spmm((n, m), ip, ix, dt, rhs)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-12 * scale
    assert np.array_equal(got[lens <= max(long_row, 1)], ref[lens <= max(long_row, 1)])
    out0 = rng.standard_normal((n, k))
    res = prod.DeviceMat.from_host(out0)
    prod.csr_mulacc_dense_rowmaj(a, prod.DeviceMat.from_host(rhs), res)
    ref2 = oracle_spmm((n, m), ip, ix, dt, rhs, out0)
    assert np.abs(res.to_host() - ref2).max() <= 1e-12 * scale
    assert np.array_equal(res.to_host()[lens == 0], out0[lens == 0])


@pytest.mark.parametrize("k", [1, 5, 8, 16, 24, 64, 70])
@pytest.mark.parametrize("idx,ptr", IDX_COMBOS[:2])
def test_stream_tiles_and_runs(hip, k, idx, ptr):
    """the entry-stream kernel at its seams: rows that start exactly at a tile (256 entries) or run (256 / G) boundary, rows that
    cover several whole tiles, a row that begins in one tile's last entry, runs of empty rows across tile boundaries and at both
    ends, a last tile with a handful of entries; the accumulate form; strided (column-major) operands"""
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat
    rng = np.random.default_rng(7 * k + np.dtype(idx).itemsize)
    lens = [0, 0, 256, 0, 32, 32, 64, 127, 1, 0, 0, 0, 255, 1, 256 * 3, 31, 1, 1000, 24, 0, 8, 8, 8, 8, 200, 56, 255, 257, 0, 0, 3]
    lens += [int(v) for v in rng.integers(0, 70, 400)] + [0] * 70 + [513, 0, 0]
    lens = np.array(lens)
    n, m = lens.size, 1500
    ip = np.zeros(n + 1, ptr)
    ip[1:] = np.cumsum(lens)
    ix = np.concatenate([np.sort(rng.choice(m, int(l), replace=False)) for l in lens]).astype(idx)
    dt = rng.standard_normal(ix.size)
    rhs = rng.standard_normal((m, k))
    a = DeviceCsMat.from_host((n, m), ip, ix, dt)
    ref = oracle_spmm((n, m), ip, ix, dt, rhs)
    bound = oracle_spmm((n, m), ip, ix, np.abs(dt), np.abs(rhs))
    for relayout in (2, 1):                                       # from the rhs itself, from its re-laid-out copy: the same bits
        hip.set_option("spmm_relayout", relayout)
        try:
            for col_major in (False, True):
                got = (a * prod.DeviceMat.from_host(rhs, col_major=col_major)).to_host()
                assert np.all(np.abs(got - ref) <= 64 * np.finfo(float).eps * bound)
                assert np.all(got[lens == 0] == 0.0)
                assert np.array_equal(got[lens == 1], ref[lens == 1])
                if relayout == 2 and not col_major:
                    plain = got
                elif not col_major:
                    assert np.array_equal(got, plain)
        finally:
            hip.set_option("spmm_relayout", 0)
    out0 = rng.standard_normal((n, k))
    ref2 = oracle_spmm((n, m), ip, ix, dt, rhs, out0)
    for col_major in (False, True):
        res = prod.DeviceMat.from_host(out0, col_major=col_major)
        kern = prod.csr_mulacc_dense_colmaj if col_major else prod.csr_mulacc_dense_rowmaj
        kern(a, prod.DeviceMat.from_host(rhs), res)
        assert np.all(np.abs(res.to_host() - ref2) <= 64 * np.finfo(float).eps * (np.abs(out0) + bound))
        assert np.array_equal(res.to_host()[lens == 0], out0[lens == 0])
    hip.set_option("spmm_stream", 0)                              # the chunk kernels agree to the summation order
    try:
        other = (a * prod.DeviceMat.from_host(rhs)).to_host()
    finally:
        hip.set_option("spmm_stream", 1)
    assert np.all(np.abs(other - ref) <= 64 * np.finfo(float).eps * bound)


def test_ragged_and_contract(hip, golden):
    from sprs_amd import SprsHipError, _ffi, prod
    from sprs_amd.device import DeviceCsMat
    lens = [0, 511, 512, 513, 0, 1, 2000, 3, 0]
    shape, ip, ix, dt = ragged_csr(lens, 5000, seed=2, positive=False)
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal((5000, 12))
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    got = (a * prod.DeviceMat.from_host(rhs)).to_host()
    ref = oracle_spmm(shape, ip, ix, dt, rhs)
    bound = oracle_spmm(shape, ip, ix, np.abs(dt), np.abs(rhs))
    assert np.all(np.abs(got - ref) <= TOL * bound)
    m = DeviceCsMat.from_host(*as_csr(golden["mat3"]))           # 5 x 4
    with pytest.raises(SprsHipError, match="Dimension mismatch") as e:      # prod.rs:199-201
        prod.csr_mulacc_dense_rowmaj(m, prod.DeviceMat(5, 3), prod.DeviceMat(5, 3))
    assert e.value.status == _ffi.DIM_MISMATCH
    with pytest.raises(SprsHipError, match="Dimension mismatch"):
        prod.csr_mulacc_dense_rowmaj(m, prod.DeviceMat(4, 3), prod.DeviceMat(5, 2))
    shape, ip, ix, dt = as_csr(golden["mat1_csc"])
    csc = DeviceCsMat.from_host(shape, ip, ix, dt, storage=_ffi.CSC)
    with pytest.raises(SprsHipError, match="Storage mismatch"):             # prod.rs:202
        prod.csr_mulacc_dense_rowmaj(csc, prod.DeviceMat(5, 2), prod.DeviceMat(5, 2))
    # both wrong: the reference asserts the dimensions first and the storage last (prod.rs:199-202), and so does every mirror
    with pytest.raises(SprsHipError, match="Dimension mismatch"):
        prod.csr_mulacc_dense_rowmaj(csc, prod.DeviceMat(4, 2), prod.DeviceMat(5, 2))
    with pytest.raises(SprsHipError, match="Dimension mismatch"):
        prod.csc_mulacc_dense_colmaj(m, prod.DeviceMat(3, 2), prod.DeviceMat(5, 2))
    # a result that lies inside the right-hand side is refused (as for the vectors of the SpMV entries)
    sq = DeviceCsMat.eye(6)
    buf = prod.DeviceMat(6, 4)
    with pytest.raises(SprsHipError) as e:
        prod.csr_mulacc_dense_rowmaj(sq, buf, buf)
    assert e.value.status == _ffi.INVALID_ARG
    # ... but two element-disjoint views of ONE buffer are legal, as ndarray views are for the reference:
    # rhs = buf[:, 0:k], out = buf[:, k:2k] with pitch 2k (row-major), and the same as column blocks (column-major)
    import ctypes as C
    from sprs_amd.device import DeviceVec
    sq2 = DeviceCsMat.from_host((6, 6), np.arange(7, dtype=np.uint64), np.array([5, 4, 3, 2, 1, 0], dtype=np.uint64),
                                np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]))          # a scaled row reversal
    for layout, ld in ((_ffi.ROW_MAJOR, 8), (_ffi.COL_MAJOR, 12)):
        host = rng.standard_normal(48)
        both = DeviceVec.from_host(host)
        off = 4 if layout == _ffi.ROW_MAJOR else 6                     # row-major: columns 4..7 of every 8-wide row; column-major: rows 6..11 of every 12-high column
        _ffi.check(_ffi.lib.sprs_hip_csmat_mulacc_dense_f64(sq2._h, C.c_void_p(both.ptr), 6, 4, layout, ld, C.c_void_p(both.ptr + off * 8), 6, layout, ld, 0, None))
        got = both.to_host()
        if layout == _ffi.ROW_MAJOR:
            v = got.reshape(6, 8)
            src, dst = host.reshape(6, 8)[:, :4], v[:, 4:]
            assert np.array_equal(v[:, :4], src)
        else:
            v = got.reshape(4, 12).T                                   # 12 x 4, column-major storage
            src, dst = host.reshape(4, 12).T[:6, :], v[6:, :]
            assert np.array_equal(v[:6, :], src)
        assert np.array_equal(dst, np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])[:, None] * src[::-1, :])
        # the same pitch with views that DO share elements is still refused
        with pytest.raises(SprsHipError) as e:
            _ffi.check(_ffi.lib.sprs_hip_csmat_mulacc_dense_f64(sq2._h, C.c_void_p(both.ptr), 6, 4, layout, ld, C.c_void_p(both.ptr + (off - 1) * 8), 6, layout, ld, 0, None))
        assert e.value.status == _ffi.INVALID_ARG


def test_hypersparse_operator_form(hip):
    """many more rows than entries: the operator form clears the result in one piece and runs the accumulate
    kernels on it — same values, every empty row +0.0 — for both result layouts (row-major from 8 columns, `.f()` below), a
    row-major and a column-major rhs, and rows at both ends of the matrix empty"""
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat
    rng = np.random.default_rng(41)
    rows, cols, nnz = 300000, 5000, 700
    r = np.sort(rng.choice(np.arange(100, rows - 100), size=nnz // 2, replace=False))
    lens = np.zeros(rows, dtype=np.int64)
    lens[r] = rng.integers(1, 4, size=r.size)
    ip = np.zeros(rows + 1, dtype=np.uint64)
    ip[1:] = np.cumsum(lens)
    ix = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens[r]]).astype(np.uint64)
    dt = rng.standard_normal(ix.size)
    a = DeviceCsMat.from_host((rows, cols), ip, ix, dt)
    for k in (3, 16):
        rhs = rng.standard_normal((cols, k))
        ref = oracle_spmm((rows, cols), ip, ix, dt, rhs)
        for col_major in (False, True):
            out = (a * prod.DeviceMat.from_host(rhs, col_major=col_major)).to_host()
            assert rel_err(out, ref) <= TOL
            assert not out[lens == 0].any() and not np.signbit(out[lens == 0]).any()


def test_dense_dispatch_below_the_abi(hip):
    """`&CsMat * &Array2` / `&CsMat * &Array1` / `Array2::dot(&CsMat)` dispatched by the LIBRARY (csmat.rs:1989-2160):
    both storages, both rhs layouts, both result layouts (row-major from 8 columns, `.f()` below), the accumulate kernels of
    prod.rs with explicit layouts, the CSC -> CSR copy cached in the handle (second product of the same handle; refresh)."""
    import ctypes as C
    import scipy.sparse as sp
    from sprs_amd import _ffi, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec, CSC
    rng = np.random.default_rng(12)
    u = lambda v: np.asarray(v, dtype=np.uint64)
    m = sp.random(700, 500, density=0.03, random_state=9, format="csr")
    m.data[:] = rng.standard_normal(m.nnz)
    m.sort_indices()
    mc = m.tocsc()
    mc.sort_indices()
    a_csr = DeviceCsMat.from_host((700, 500), u(m.indptr), u(m.indices), m.data)
    a_csc = DeviceCsMat.from_host((700, 500), u(mc.indptr), u(mc.indices), mc.data, storage=CSC)
    for a in (a_csr, a_csc):
        for k in (1, 3, 7, 8, 20):
            rhs = rng.standard_normal((500, k))
            ref = oracle_spmm((700, 500), u(m.indptr), u(m.indices), m.data, rhs)
            for rhs_col_major in (False, True):
                out = a * prod.DeviceMat.from_host(rhs, col_major=rhs_col_major)
                assert out.col_major == (k < 8)                       # csmat.rs:2002-2045: `.f()` below 8 columns
                assert rel_err(out.to_host(), ref) <= TOL
                for out_col_major in (False, True):                   # the accumulate kernels, every layout pair
                    out0 = rng.standard_normal((700, k))
                    acc = prod.DeviceMat.from_host(out0, col_major=out_col_major)
                    kernel = {(False, False): prod.csr_mulacc_dense_rowmaj, (False, True): prod.csr_mulacc_dense_colmaj,
                              (True, False): prod.csc_mulacc_dense_rowmaj, (True, True): prod.csc_mulacc_dense_colmaj}[(a.is_csc(), out_col_major)]
                    kernel(a, prod.DeviceMat.from_host(rhs, col_major=rhs_col_major), acc)
                    ref_acc = oracle_spmm((700, 500), u(m.indptr), u(m.indices), m.data, rhs, out0)
                    assert rel_err(acc.to_host(), ref_acc) <= TOL
        x = rng.standard_normal(500)
        y = (a * DeviceVec.from_host(x)).to_host()                    # `&A * &x`, CSC included (csmat.rs:2140-2156)
        assert rel_err(y, m @ x) <= TOL
    # prod::mul_acc_mat_vec_csc and its contract (prod.rs:88-92)
    y0 = rng.standard_normal(700)
    yv = DeviceVec.from_host(y0)
    prod.mul_acc_mat_vec_csc(a_csc, DeviceVec.from_host(x), yv)
    assert rel_err(yv.to_host(), y0 + m @ x) <= TOL
    with pytest.raises(hip.SprsHipError) as e:
        prod.mul_acc_mat_vec_csc(a_csr, DeviceVec.from_host(x), yv)
    assert e.value.status == _ffi.STORAGE_MISMATCH
    with pytest.raises(hip.SprsHipError) as e:
        prod.mul_acc_mat_vec_csc(a_csc, DeviceVec.from_host(np.zeros(499)), yv)
    assert e.value.status == _ffi.DIM_MISMATCH
    with pytest.raises(hip.SprsHipError) as e:                         # the twin of mul_acc_mat_vec_csr keeps ITS storage assert
        prod.mul_acc_mat_vec_csr(a_csc, DeviceVec.from_host(x), yv)
    assert e.value.status == _ffi.STORAGE_MISMATCH
    # dense . sparse (csmat.rs:2050-2117), both storages of the sparse operand, both layouts of the dense one
    for a, ms in ((a_csr, m), (a_csc, m)):
        for k in (3, 9):
            lhs = rng.standard_normal((k, 700))
            for cm in (False, True):
                got = prod.dense_dot_csmat(prod.DeviceMat.from_host(lhs, col_major=cm), a).to_host()
                assert got.shape == (k, 500) and rel_err(got, lhs @ ms.toarray()) <= 1e-9
    # a wrapped CSC handle whose values change in place: refresh drops the cached CSR copy
    import torch
    if not torch.cuda.is_available():                                  # (the CPU kernel emulator runs this test up to here)
        return
    dev = torch.device("cuda", 0)
    ip = torch.from_numpy(mc.indptr.astype(np.int64)).to(dev)
    ix = torch.from_numpy(mc.indices.astype(np.int64)).to(dev)
    dt = torch.from_numpy(mc.data.copy()).to(dev)
    w = DeviceCsMat.wrap_torch((700, 500), ip, ix, dt, storage=CSC)
    xv = DeviceVec.from_host(x)
    assert rel_err((w * xv).to_host(), m @ x) <= TOL
    dt.mul_(2.0)
    torch.cuda.synchronize()
    _ffi.check(_ffi.lib.sprs_hip_csmat_refresh(w._h))
    assert rel_err((w * xv).to_host(), 2.0 * (m @ x)) <= TOL
