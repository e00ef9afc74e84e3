"""GPU parity of the triplet -> CSR / CSC assembly (sprs_amd/triplet.py, twin of TriMatBase::to_csr /
to_csc, triplet.rs:262-276 -> triplet_iter.rs:127-224) against the oracle's restatement: structure
bit-exact; values bit-exact too, because both fold duplicates in triplet order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import sprs_amd
    if sprs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (no CPU fallback exists)")
    return sprs_amd


def random_triplets(rows, cols, n, seed, hot=0.2):
    rng = np.random.default_rng(seed)
    r = rng.integers(0, rows, n)
    c = rng.integers(0, cols, n)
    m = rng.random(n) < hot                                   # a few cells collect many duplicates
    r[m] = rng.integers(0, 3, m.sum())
    c[m] = rng.integers(0, 3, m.sum())
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)  # order of the additions is visible in the bits
    return r, c, v


@pytest.mark.parametrize("idx", [np.uint64, np.uint32])
@pytest.mark.parametrize("storage", ["CSR", "CSC"])
def test_assembly_matches_oracle(hip, storage, idx):
    from oracle import oracle
    from sprs_amd.triplet import TriMat
    rows, cols, n = 700, 450, 60000
    r, c, v = random_triplets(rows, cols, n, seed=3)
    t = TriMat((rows, cols), r, c, v)
    m = t.to_csr(idx) if storage == "CSR" else t.to_csc(idx)
    shape, ip, ix, dt = m.to_host()
    rip, rix, rdt = oracle.triplets_to_cs((rows, cols), r, c, v, storage=storage, idx_dtype=idx)
    assert shape == (rows, cols) and (m.is_csr() if storage == "CSR" else m.is_csc())
    assert ix.dtype == np.dtype(idx)
    assert np.array_equal(ip, rip) and np.array_equal(ix, rix)
    assert np.array_equal(dt, rdt)
    assert int(rip[-1]) < n                                   # duplicates were folded


def test_zeros_cancellation_empty_and_big_rows(hip):
    from oracle import oracle
    from sprs_amd.triplet import TriMat
    # explicit zero, a sum that cancels (stays stored), an empty row, an empty matrix
    t = TriMat((4, 4), [2, 0, 2, 0, 2, 1], [1, 3, 1, 0, 1, 2], [1e16, 5.0, 1.0, 0.0, -1e16, 7.0])
    shape, ip, ix, dt = t.to_csr().to_host()
    rip, rix, rdt = oracle.triplets_to_cs((4, 4), t.row_inds, t.col_inds, t.data)
    assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(dt, rdt)
    assert dt.tolist() == [0.0, 5.0, 7.0, (1e16 + 1.0) + -1e16]
    e = TriMat((3, 2))
    assert e.to_csr().to_host()[1].tolist() == [0, 0, 0, 0] and e.to_csc().to_host()[1].tolist() == [0, 0, 0]
    # one row collecting 5000 triplets over 40 columns (large-row SpGEMM path of the selector product)
    rng = np.random.default_rng(1)
    n = 5000
    r = np.full(n, 7)
    c = rng.integers(0, 40, n)
    v = rng.standard_normal(n)
    t = TriMat((9, 40), r, c, v)
    shape, ip, ix, dt = t.to_csr().to_host()
    rip, rix, rdt = oracle.triplets_to_cs((9, 40), r, c, v)
    assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(dt, rdt)


def test_matrix_market_to_device_and_multiply(hip):
    """read -> assemble on the device -> SpMV, against the dense matrix scipy reads from the same text"""
    import io
    import scipy.io
    from sprs_amd.device import DeviceVec
    from sprs_amd.io import read_matrix_market, write_matrix_market
    from sprs_amd.triplet import TriMat
    r, c, v = random_triplets(300, 300, 8000, seed=9, hot=0.05)
    buf = io.StringIO()
    write_matrix_market(buf, TriMat((300, 300), r, c, v))
    text = buf.getvalue()
    a = read_matrix_market(io.StringIO(text)).to_csr()
    dense = scipy.io.mmread(io.BytesIO(text.encode())).toarray()
    x = np.random.default_rng(2).standard_normal(300)
    y = (a * DeviceVec.from_host(x)).to_host()
    assert np.allclose(y, dense @ x, rtol=1e-10, atol=1e-8)
    # and back out through the writer: same matrix
    out = io.StringIO()
    write_matrix_market(out, a)
    assert np.allclose(scipy.io.mmread(io.BytesIO(out.getvalue().encode())).toarray(), dense, rtol=1e-13, atol=1e-13)


def test_reference_golden_cases(hip, golden):
    """the reference's own triplet tests (triplet.rs:343-646, tests/golden/sprs_fixtures.json): to_csc must give the CSC
    it asserts, to_csr that matrix converted (`expected.to_csr()`), for 8- and 4-byte indices"""
    from oracle import oracle
    from sprs_amd.triplet import TriMat
    for c in golden["triplet_cases"]:
        rows, cols = c["shape"]
        exp = c["csc"]
        eip, eix, edt = oracle.convert_storage(cols, rows, np.array(exp["indptr"], dtype=np.uint64),
                                               np.array(exp["indices"], dtype=np.uint64), np.array(exp["data"]))
        for idx in (np.uint64, np.uint32):
            t = TriMat((rows, cols), c["rows"], c["cols"], c["data"])
            shape, ip, ix, dt = t.to_csc(idx).to_host()
            assert shape == (rows, cols)
            assert ip.tolist() == exp["indptr"] and ix.tolist() == exp["indices"] and dt.tolist() == exp["data"], c["name"]
            shape, ip, ix, dt = t.to_csr(idx).to_host()
            assert ip.tolist() == eip.tolist() and ix.tolist() == eix.tolist() and dt.tolist() == edt.tolist(), c["name"]


@pytest.mark.parametrize("storage", ["CSR", "CSC"])
def test_sort_route_equals_selector_product(hip, storage):
    """the radix-sort kernel route (sprs_hip_triplets_to_cs) and the selector-product route (to_other_storage +
    mul_csr_csr) fold duplicates in the same (triplet) order: identical bits; 2e5 triplets with heavy duplication,
    indices that need more than one radix pass in both fields"""
    from sprs_amd.triplet import TriMat
    rows, cols, n = 70000, 300, 200000
    r, c, v = random_triplets(rows, cols, n, seed=11, hot=0.3)
    t = TriMat((rows, cols), r, c, v)
    a = (t.to_csr() if storage == "CSR" else t.to_csc()).to_host()
    b = (t.to_csr(method="product") if storage == "CSR" else t.to_csc(method="product")).to_host()
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)


def test_out_of_bounds_and_widths(hip):
    import ctypes as C
    from sprs_amd import _ffi
    from sprs_amd.device import DeviceVec
    r = DeviceVec.from_host(np.array([0, 5], dtype=np.uint64).view(np.float64))
    c = DeviceVec.from_host(np.array([1, 1], dtype=np.uint64).view(np.float64))
    v = DeviceVec.from_host(np.array([1.0, 2.0]))
    h = C.c_void_p()
    st = _ffi.lib.sprs_hip_triplets_to_cs(4, 4, 2, C.c_void_p(r.ptr), C.c_void_p(c.ptr), 8, C.c_void_p(v.ptr), 0, 8, 8, C.byref(h))
    assert st == _ffi.INVALID_ARG and b"out of bounds" in _ffi.lib.sprs_hip_last_error()
    st = _ffi.lib.sprs_hip_triplets_to_cs(4, 4, 2, C.c_void_p(r.ptr), C.c_void_p(c.ptr), 2, C.c_void_p(v.ptr), 0, 8, 8, C.byref(h))
    assert st == _ffi.INVALID_ARG
