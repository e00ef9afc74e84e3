"""GPU parity tests of the device Gauss-Seidel (twin of gauss_seidel() of the reference's heat example,
sprs/examples/heat.rs:103-139) against the CPU oracle's restatement: every iterate bit for bit, the
convergence scalar to 1e-10 of sum |r_i|, the iteration counts, the reference's panics as errors.
(tests/test_emu_cpu.py runs this file through the CPU emulator of the kernels as well.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EMU = "emu" in os.path.basename(os.environ.get("SPRS_HIP_LIBRARY", ""))


@pytest.fixture(scope="module")
def hip():
    import sprs_amd
    if sprs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (no CPU fallback exists)")
    return sprs_amd


def heat_system(rows, idx=np.uint64, ptr=np.uint64):
    """main() of heat.rs:141-157: grid_laplacian((rows, rows)), rhs = row + col on the border, 0 inside"""
    from oracle import oracle
    shape, ip, ix, dt = oracle.grid_laplacian(rows, rows, idx_dtype=idx, ptr_dtype=ptr)
    i, j = np.meshgrid(np.arange(rows), np.arange(rows), indexing="ij")
    border = (i == 0) | (i == rows - 1) | (j == 0) | (j == rows - 1)
    rhs = np.where(border, (i + j).astype(np.float64), 0.0).reshape(-1)
    return shape, ip, ix, dt, rhs


def gpu_gs(shape, ip, ix, dt, x0, rhs, max_iter, eps):
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.linalg import gauss_seidel
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    x = DeviceVec.from_host(x0)
    res = gauss_seidel(a, x, DeviceVec.from_host(rhs), max_iter, eps)
    return x.to_host(), res


def same_error(e_gpu, e_ref, scale):
    """the convergence scalar sqrt(sum r_i): NaN where the reference's is (negative sum), else to 1e-10 of sum |r_i|"""
    if np.isnan(e_ref):
        return np.isnan(e_gpu)
    return abs(e_gpu * e_gpu - e_ref * e_ref) <= 1e-10 * max(scale, 1e-300)


@pytest.mark.parametrize("idx,ptr", [(np.uint64, np.uint64), (np.uint32, np.uint64), (np.uint32, np.uint32)])
def test_heat_example(hip, idx, ptr):
    """the example itself (heat.rs:141-157): 10 x 10 grid, 300 sweeps, eps 1e-8 — converges in the same sweep as the CPU
    restatement, to the same bits, and to the solution of the linear system"""
    from oracle import oracle
    shape, ip, ix, dt, rhs = heat_system(10, idx, ptr)
    x_ref, info = oracle.gauss_seidel(shape, ip, ix, dt, np.zeros(100), rhs, 300, 1e-8)
    x, res = gpu_gs(shape, ip, ix, dt, np.zeros(100), rhs, 300, 1e-8)
    assert res.converged and info["converged"] == 1
    assert res.iterations == info["iterations"]
    assert np.array_equal(x, x_ref)
    assert same_error(res.error, info["error"], 1.0)
    assert res.levels == 16                                     # border rows: level 0; interior (i, j): level i + j - 1 <= 15
    import scipy.sparse as sp
    dense = sp.csr_matrix((dt, ix.astype(np.int64), ip.astype(np.int64)), shape=shape).toarray()
    assert np.abs(x - np.linalg.solve(dense, rhs)).max() < 1e-12


@pytest.mark.parametrize("sweeps", [1, 2, 7])
def test_every_iterate_bit_for_bit(hip, sweeps):
    """a grid with more rows than one wave chunk and more levels than waves in a workgroup; Err(error) after k sweeps"""
    from oracle import oracle
    rows = 24 if EMU else 96
    shape, ip, ix, dt, rhs = heat_system(rows)
    x0 = np.random.default_rng(3).standard_normal(rows * rows)
    x_ref, info = oracle.gauss_seidel(shape, ip, ix, dt, x0, rhs, sweeps, -1.0)
    x, res = gpu_gs(shape, ip, ix, dt, x0, rhs, sweeps, -1.0)
    assert not res.converged and info["converged"] == 0 and res.iterations == sweeps == info["iterations"]
    assert np.array_equal(x, x_ref)
    import scipy.sparse as sp
    a = sp.csr_matrix((dt, ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    assert same_error(res.error, info["error"], np.abs(a @ x_ref - rhs).sum())
    assert res.levels == 2 * rows - 4


def _random_system(n, seed, density, long_row=0):
    """non-symmetric, strictly diagonally dominant, unsorted positions of the diagonal inside the rows; `long_row` entries
    in row n // 2 (more than one batch of eight, dependencies on both sides of the diagonal)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), density)
    c = rng.integers(0, n, size=n * density)
    if long_row:
        r = np.concatenate([r, np.full(long_row, n // 2)])
        c = np.concatenate([c, rng.integers(0, n, size=long_row)])
    a = sp.coo_matrix((rng.standard_normal(r.size), (r, c)), shape=(n, n)).tocsr()
    a = (a + sp.diags(np.abs(a).sum(axis=1).A1 + 1.0 + rng.random(n))).tocsr()
    a.sort_indices()
    return a


@pytest.mark.parametrize("n,density,long_row", [(1, 0, 0), (63, 3, 0), (64, 2, 20), (500, 4, 100), (3000, 6, 400)])
def test_random_nonsymmetric_systems(hip, n, density, long_row):
    """rows whose columns before the diagonal are NOT mirrored by entries after it (the previous iterate must stay readable
    while later rows are already swept), rows of many entries, a single row"""
    from oracle import oracle
    if EMU and n > 600:
        pytest.skip("emulator: small systems only")
    a = _random_system(n, 11 + n, density, long_row)
    ip, ix, dt = a.indptr.astype(np.uint64), a.indices.astype(np.uint64), a.data
    rng = np.random.default_rng(n)
    rhs, x0 = rng.standard_normal(n), rng.standard_normal(n)
    for sweeps in (1, 3):
        x_ref, info = oracle.gauss_seidel((n, n), ip, ix, dt, x0, rhs, sweeps, -1.0)
        x, res = gpu_gs((n, n), ip, ix, dt, x0, rhs, sweeps, -1.0)
        assert np.array_equal(x, x_ref)
        assert same_error(res.error, info["error"], np.abs(a @ x_ref - rhs).sum())
    # and to convergence: the same sweep count (eps far from the rounding of the scalar)
    x_ref, info = oracle.gauss_seidel((n, n), ip, ix, dt, x0, rhs, 200, 1e-6)
    x, res = gpu_gs((n, n), ip, ix, dt, x0, rhs, 200, 1e-6)
    assert (res.converged, res.iterations) == (bool(info["converged"]), info["iterations"])
    assert np.array_equal(x, x_ref)


def test_lower_triangular_chain(hip):
    """bidiagonal matrix: every row waits for the one before it (n levels, all of them inside / across waves)"""
    from oracle import oracle
    import scipy.sparse as sp
    n = 200 if EMU else 1000
    a = (sp.diags(np.full(n, 2.0)) + sp.diags(np.full(n - 1, -1.0), -1)).tocsr()
    ip, ix, dt = a.indptr.astype(np.uint64), a.indices.astype(np.uint32), a.data
    rhs = np.arange(1, n + 1, dtype=np.float64)
    x_ref, info = oracle.gauss_seidel((n, n), ip, ix, dt, np.zeros(n), rhs, 2, 1e-9)
    x, res = gpu_gs((n, n), ip, ix, dt, np.zeros(n), rhs, 2, 1e-9)
    assert res.levels == n
    assert np.array_equal(x, x_ref)
    assert (res.converged, res.iterations) == (bool(info["converged"]), info["iterations"])


def test_zero_sweeps_returns_the_initial_error(hip):
    """max_iter = 0: Err(error of the start vector) (heat.rs:111, 138), x untouched"""
    from oracle import oracle
    shape, ip, ix, dt, rhs = heat_system(8)
    x0 = np.linspace(-1, 2, 64)
    x_ref, info = oracle.gauss_seidel(shape, ip, ix, dt, x0, rhs, 0, 1e-8)
    x, res = gpu_gs(shape, ip, ix, dt, x0, rhs, 0, 1e-8)
    assert np.array_equal(x, x0) and np.array_equal(x_ref, x0)
    assert not res.converged and res.iterations == 0
    import scipy.sparse as sp
    a = sp.csr_matrix((dt, ix.astype(np.int64), ip.astype(np.int64)), shape=shape)
    assert same_error(res.error, info["error"], np.abs(a @ x0 - rhs).sum())


def test_nan_and_inf_pass_through(hip):
    """a zero diagonal divides by zero like the reference (inf / NaN iterates, no error); a NaN with every payload bit set in
    rhs must not be mistaken for 'not swept yet'"""
    from oracle import oracle
    import scipy.sparse as sp
    n = 70
    a = (sp.diags(np.full(n, 3.0)) + sp.diags(np.full(n - 1, 1.0), -1) + sp.diags(np.full(n - 2, 0.5), 2)).tocsr()
    a.sort_indices()
    ip, ix, dt = a.indptr.astype(np.uint64), a.indices.astype(np.uint64), a.data.copy()
    (p5,) = [p for p in range(int(ip[5]), int(ip[6])) if ix[p] == 5]
    dt[p5] = 0.0                                               # a STORED zero on the diagonal of row 5
    rhs = np.ones(n)
    rhs[40] = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFFFF).tobytes(), dtype=np.float64)[0]
    x_ref, info = oracle.gauss_seidel((n, n), ip, ix, dt, np.zeros(n), rhs, 2, 1e-9)
    x, res = gpu_gs((n, n), ip, ix, dt, np.zeros(n), rhs, 2, 1e-9)
    assert np.array_equal(np.isnan(x), np.isnan(x_ref))
    ok = ~np.isnan(x_ref)
    assert np.array_equal(x[ok], x_ref[ok])
    assert not res.converged and info["converged"] == 0


def test_contract_violations(hip):
    """heat.rs:109-110 asserts, heat.rs:127 diag.unwrap(), CSC operand"""
    from sprs_amd import _ffi
    from sprs_amd.device import DeviceCsMat, DeviceVec, CSC
    from sprs_amd.linalg import gauss_seidel
    from oracle import oracle
    shape, ip, ix, dt, rhs = heat_system(6)
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    with pytest.raises(_ffi.SprsHipError) as e:
        gauss_seidel(a, DeviceVec.zeros(35), DeviceVec.zeros(35), 3, 1e-8)
    assert e.value.status == _ffi.DIM_MISMATCH
    rect = DeviceCsMat.from_host((2, 3), np.array([0, 1, 2], dtype=np.uint64), np.array([0, 1], dtype=np.uint64), np.ones(2))
    with pytest.raises(_ffi.SprsHipError) as e:
        gauss_seidel(rect, DeviceVec.zeros(2), DeviceVec.zeros(2), 3, 1e-8)
    assert e.value.status == _ffi.DIM_MISMATCH
    csc = DeviceCsMat.from_host(shape, ip, ix, dt, storage=CSC)
    with pytest.raises(_ffi.SprsHipError) as e:
        gauss_seidel(csc, DeviceVec.zeros(36), DeviceVec.zeros(36), 3, 1e-8)
    assert e.value.status == _ffi.STORAGE_MISMATCH
    # a row without a diagonal entry: the reference panics in the first sweep; with no sweep asked for it does not
    nd_ip = np.array([0, 1, 2, 3], dtype=np.uint64)
    nd_ix = np.array([0, 0, 2], dtype=np.uint64)
    nd = DeviceCsMat.from_host((3, 3), nd_ip, nd_ix, np.ones(3))
    with pytest.raises(_ffi.SprsHipError) as e:
        gauss_seidel(nd, DeviceVec.zeros(3), DeviceVec.zeros(3), 1, 1e-8)
    assert e.value.status == _ffi.BAD_STRUCTURE and "row 1" in str(e.value)
    with pytest.raises(oracle.OracleError):
        oracle.gauss_seidel((3, 3), nd_ip, nd_ix, np.ones(3), np.zeros(3), np.zeros(3), 1, 1e-8)
    res = gauss_seidel(nd, DeviceVec.zeros(3), DeviceVec.zeros(3), 0, 1e-8)
    assert not res.converged and res.error == 0.0


def test_plan_is_kept_and_follows_the_handle(hip):
    """two solves on one handle (level order built once) give the same bits; a second handle with other structure its own"""
    shape, ip, ix, dt, rhs = heat_system(12)
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.linalg import gauss_seidel
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    outs = []
    for _ in range(2):
        x = DeviceVec.zeros(144)
        r = gauss_seidel(a, x, DeviceVec.from_host(rhs), 5, -1.0)
        outs.append(x.to_host())
        assert r.levels == 20
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("rows,sweeps", [(64, 1), (80, 3), (130, 2)])
def test_band_schedule_bit_for_bit(hip, rows, sweeps):
    """option gauss_seidel_chain = S (the grid width): chains of S consecutive rows per lane, skewed, hand-offs through LDS inside a
    workgroup — 64 x 64 is one band, 80 x 80 two bands (dependencies across the band edge polled in memory), 130 x 130 a band whose
    chains end inside it.  Every iterate bit for bit, like the level-order kernel; a stride the matrix does not fit is refused."""
    from oracle import oracle
    if EMU and rows > 80:
        rows, sweeps = 66, 2
    shape, ip, ix, dt, rhs = heat_system(rows)
    x0 = np.random.default_rng(5).standard_normal(rows * rows)
    x_ref, info = oracle.gauss_seidel(shape, ip, ix, dt, x0, rhs, sweeps, -1.0)
    hip.set_option("gauss_seidel_chain", rows)
    try:
        x, res = gpu_gs(shape, ip, ix, dt, x0, rhs, sweeps, -1.0)
    finally:
        hip.set_option("gauss_seidel_chain", 0)
    assert res.iterations == sweeps and np.array_equal(x, x_ref)
    x1, _ = gpu_gs(shape, ip, ix, dt, x0, rhs, sweeps, -1.0)           # level order (auto: the grid is too narrow for a band)
    assert np.array_equal(x1, x_ref)
    hip.set_option("gauss_seidel_chain", rows - 1)                        # row i - 1 of a chain's first row is swept later: refused
    try:
        with pytest.raises(hip.SprsHipError) as e:
            gpu_gs(shape, ip, ix, dt, x0, rhs, 1, -1.0)
        assert e.value.status == hip._ffi.INVALID_ARG
    finally:
        hip.set_option("gauss_seidel_chain", 0)
