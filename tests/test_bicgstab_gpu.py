"""GPU parity tests of the device-resident BiCGSTAB (twin of sprs::linalg::bicgstab)
against the CPU oracle's restatement and the reference's own example system."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import sprs_amd
    if sprs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (no CPU fallback exists)")
    return sprs_amd


def gpu_solve(shape, ip, ix, dt, x0, b, tol, max_iter, storage="CSR", thr=0.1):
    from sprs_amd.device import DeviceCsMat, DeviceVec, CSR, CSC
    from sprs_amd.linalg import BiCGSTAB
    a = DeviceCsMat.from_host(shape, ip, ix, dt, storage=CSR if storage == "CSR" else CSC)
    return BiCGSTAB.solve(a, DeviceVec.from_host(x0), DeviceVec.from_host(b), tol, max_iter, thr)


@pytest.mark.parametrize("idx", [np.uint64, np.uint32])
def test_reference_example_system(hip, golden, idx):
    """bicgstab.rs:32-66 / 336-369: CSC 4x4, b = x0 = ones, tol 1e-60, 50 iterations must converge and
    A x must equal b to 1e-60 relative, i.e. exactly.  n <= 2048 runs the serial dot: same 45 iterations
    and restarts as the oracle, same bits in x."""
    from oracle import oracle
    fx = golden["bicgstab_example"]
    ip, ix = np.array(fx["indptr"], dtype=idx), np.array(fx["indices"], dtype=idx)
    dt = np.array(fx["data"])
    res = gpu_solve((4, 4), ip, ix, dt, np.ones(4), np.ones(4), fx["tol"], fx["max_iter"], storage="CSC")
    x_ref, info = oracle.bicgstab((4, 4), ip, ix, dt, np.ones(4), np.ones(4), fx["tol"], fx["max_iter"], storage="CSC")
    assert res.converged and info["converged"] == 1
    x = res.x().to_host()
    assert np.array_equal(x, x_ref)
    assert (res.iteration_count(), res.soft_restart_count(), res.hard_restart_count()) == \
        (info["iteration_count"], info["soft_restart_count"], info["hard_restart_count"])
    assert res.err() == info["err"] == 0.0
    dense = np.zeros((4, 4))
    for j in range(4):
        for p in range(int(ip[j]), int(ip[j + 1])):
            dense[int(ix[p]), j] = dt[p]
    assert np.all(np.abs(1.0 - np.ones(4) / (dense @ x)) < fx["tol"])      # the reference test's own assertion


def _diag_dominant(n, seed, density=8):
    """random sparse, strictly diagonally dominant, non-symmetric: BiCGSTAB converges quickly.
    (numpy + coo: scipy.sparse.random needs minutes for n = 60000)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), density)
    cols = rng.integers(0, n, size=n * density)
    a = sp.coo_matrix((rng.standard_normal(n * density), (rows, cols)), shape=(n, n)).tocsr()   # duplicates summed
    a = (a + sp.diags(np.abs(a).sum(axis=1).A1 + 1.0 + rng.random(n))).tocsr()
    a.sort_indices()
    return a


def test_serial_dot_range_tracks_oracle(hip):
    """n <= 2048: dots and element-wise steps are the reference's own operations; the SpMV is not (a row
    that straddles two nnz tiles is summed as tail + head, within the SpMV's 1e-10 bar), so iterates agree
    with the oracle to rounding — same iteration and restart counts on this well-conditioned system"""
    from oracle import oracle
    n = 1500
    a = _diag_dominant(n, 3)
    u = lambda v: v.astype(np.uint64)
    rng = np.random.default_rng(1)
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    for tol, iters in ((1e-10, 60), (1e-300, 7)):          # converging run, and one cut off by max_iter
        res = gpu_solve((n, n), u(a.indptr), u(a.indices), a.data, x0, b, tol, iters)
        x_ref, info = oracle.bicgstab((n, n), u(a.indptr), u(a.indices), a.data, x0, b, tol, iters)
        assert rel_err(res.x().to_host(), x_ref) <= 1e-10
        assert res.converged == bool(info["converged"])
        assert (res.iteration_count(), res.soft_restart_count(), res.hard_restart_count()) == \
            (info["iteration_count"], info["soft_restart_count"], info["hard_restart_count"])
        # the residual norm at convergence is rounding noise of the iterates: compare loosely
        assert abs(res.err() - info["err"]) <= 1e-3 * info["err"] + 1e-15


@pytest.mark.parametrize("storage", ["CSR", "CSC"])
def test_tree_dot_range_solves_and_tracks_oracle(hip, storage):
    """n > 2048: dots are a fixed tree (rounded differently from the serial sum): the solution satisfies
    the system to the tolerance, agrees with the oracle's to 1e-10, and is bit-identical run to run"""
    from oracle import oracle
    n = 60000
    a = _diag_dominant(n, 5)
    m = a if storage == "CSR" else a.tocsc()
    m.sort_indices()
    u = lambda v: v.astype(np.uint64)
    rng = np.random.default_rng(2)
    b, x0 = rng.standard_normal(n), np.zeros(n)
    tol = 1e-9
    res = gpu_solve((n, n), u(m.indptr), u(m.indices), m.data, x0, b, tol, 200, storage=storage)
    assert res.converged and res.err() < tol and res.hard_restart_count() >= 1
    x = res.x().to_host()
    assert np.linalg.norm(a @ x - b) < tol * 1.0001
    x_ref, info = oracle.bicgstab((n, n), u(m.indptr), u(m.indices), m.data, x0, b, tol, 200, storage=storage)
    assert info["converged"] == 1 and rel_err(x, x_ref) <= 1e-10
    assert abs(res.iteration_count() - info["iteration_count"]) <= 2
    again = gpu_solve((n, n), u(m.indptr), u(m.indices), m.data, x0, b, tol, 200, storage=storage)
    assert np.array_equal(again.x().to_host(), x) and again.iteration_count() == res.iteration_count()


def test_contract_and_breakdown(hip):
    from oracle import oracle
    from sprs_amd import _ffi
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.linalg import BiCGSTAB
    a = DeviceCsMat.eye(5)
    with pytest.raises(_ffi.SprsHipError) as e:
        BiCGSTAB.solve(a, DeviceVec.zeros(4), DeviceVec.zeros(4), 1e-9, 10)
    assert e.value.status == _ffi.DIM_MISMATCH
    # the reference's breakdown, reproduced: with A = I the half step already solves the system, s = 0,
    # omega = 0 / 0 (bicgstab.rs:205) and everything turns NaN; solve() runs out of iterations -> Err
    res = BiCGSTAB.solve(a, DeviceVec.zeros(5), DeviceVec.from_host(np.arange(5.0)), 1e-12, 10)
    ip, ix = np.arange(6, dtype=np.uint64), np.arange(5, dtype=np.uint64)
    x_ref, info = oracle.bicgstab((5, 5), ip, ix, np.ones(5), np.zeros(5), np.arange(5.0), 1e-12, 10)
    assert not res.converged and info["converged"] == 0 and res.iteration_count() == info["iteration_count"] == 10
    assert np.isnan(res.x().to_host()).all() and np.isnan(x_ref).all() and np.isnan(res.err())
    # a system it can solve: 2 I x = b
    a2 = DeviceCsMat.from_host((5, 5), ip, ix, np.full(5, 2.0))
    res = BiCGSTAB.solve(a2, DeviceVec.from_host(np.ones(5)), DeviceVec.from_host(np.arange(5.0)), 1e-12, 10)
    x_ref, info = oracle.bicgstab((5, 5), ip, ix, np.full(5, 2.0), np.ones(5), np.arange(5.0), 1e-12, 10)
    assert res.converged == bool(info["converged"]) and np.array_equal(res.x().to_host(), x_ref, equal_nan=True)
