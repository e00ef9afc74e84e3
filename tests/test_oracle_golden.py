"""Pins the CPU oracle (oracle/) against the reference's own golden vectors
and known-answer tests.  CPU only."""
import numpy as np
import pytest

from conftest import IDX_COMBOS, as_csr
from oracle import oracle


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_mul_csr_vec(golden, idx, ptr):
    # sprs/src/sparse/prod.rs:375-398 (includes an empty row)
    fx = golden["mul_csr_vec"]
    shape, ip, ix, dt = as_csr(fx, idx, ptr)
    y = np.zeros(5)
    oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, np.array(fx["x"]), y)
    assert np.all(np.abs(y - np.array(fx["expected"])) < fx["epsilon"])
    assert y[1] == 0.0
    # accumulate semantics (prod.rs:121: starts from the existing y)
    y2 = np.arange(5, dtype=np.float64)
    oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, np.array(fx["x"]), y2)
    assert np.all(np.abs(y2 - (np.arange(5) + np.array(fx["expected"]))) < fx["epsilon"])
    assert y2[1] == 1.0


def test_mul_csc_vec_via_transpose(golden):
    # prod.rs:325-348: a CSC matrix is the CSR form of its transpose; convert
    # (csmat.rs:1782-1829) and use the CSR kernel.
    fx = golden["mul_csc_vec"]
    shape, ip, ix, dt = as_csr(fx)
    ip2, ix2, dt2 = oracle.convert_storage(5, 5, ip, ix, dt)
    y = np.zeros(5)
    oracle.mul_acc_mat_vec_csr(shape, ip2, ix2, dt2, np.array(fx["x"]), y)
    assert np.all(np.abs(y - np.array(fx["expected"])) < fx["epsilon"])


def test_spmv_dim_mismatch(golden):
    shape, ip, ix, dt = as_csr(golden["mat3"])  # 5x4
    with pytest.raises(oracle.OracleError) as e:
        oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, np.zeros(5), np.zeros(5))
    assert e.value.code == oracle.DIM_MISMATCH and "Dimension mismatch" in str(e.value)


def test_spmv_omp_equals_serial():
    rng = np.random.default_rng(0)
    n, m = 3000, 2500
    import scipy.sparse as sp
    a = sp.random(n, m, density=0.01, random_state=1, format="csr")
    a.sort_indices()
    x = rng.random(m)
    y1, y2 = rng.random(n), None
    y2 = y1.copy()
    args = ((n, m), a.indptr.astype(np.uint64), a.indices.astype(np.uint64), a.data)
    oracle.mul_acc_mat_vec_csr(*args, x, y1)
    oracle.mul_acc_mat_vec_csr(*args, x, y2, threads=4)
    assert np.array_equal(y1, y2)


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_symbolic_and_numeric(golden, idx, ptr):
    # smmp.rs:422-465
    a, b, exp = (as_csr(golden[k], idx, ptr) for k in ("mat1", "mat2", "mat1_matprod_mat2"))
    c_ip, c_ix = oracle.symbolic(a[0], a[1], a[2], b[0], b[1], b[2])
    assert np.array_equal(c_ip, exp[1]) and np.array_equal(c_ix, exp[2])
    c_dt = oracle.numeric(*a, *b, c_ip, c_ix)
    assert np.array_equal(c_dt, exp[3])


@pytest.mark.parametrize("threads", [0, 1, 2, 4, 8])
@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_mul_csr_csr(golden, idx, ptr, threads):
    # prod.rs:425-436, smmp.rs:467-473, 491-501 (thread-count invariance)
    a, b = as_csr(golden["mat1"], idx, ptr), as_csr(golden["mat2"], idx, ptr)
    for rhs, key in ((a, "mat1_self_matprod"), (b, "mat1_matprod_mat2")):
        exp = as_csr(golden[key], idx, ptr)
        shape, ip, ix, dt = oracle.mul_csr_csr(*a, *rhs, threads=threads)
        assert shape == exp[0]
        assert ip.dtype == ptr and ix.dtype == idx
        assert np.array_equal(ip, exp[1]) and np.array_equal(ix, exp[2]) and np.array_equal(dt, exp[3])


def test_mul_storage_dispatch(golden):
    # prod.rs:438-458 (mul_csc_csc, mul_csc_csr); dispatch csmat.rs:1933-1948
    def m(k):
        shape, ip, ix, dt = as_csr(golden[k])
        return dict(storage=golden[k]["storage"], shape=shape, indptr=ip, indices=ix, data=dt)

    def same(res, k):
        e = m(k)
        return (res["storage"] == e["storage"] and tuple(res["shape"]) == tuple(e["shape"])
                and np.array_equal(res["indptr"], e["indptr"])
                and np.array_equal(res["indices"], e["indices"])
                and np.array_equal(res["data"], e["data"]))

    assert same(oracle.csmat_mul_csmat(m("mat1_csc"), m("mat4")), "mat1_csc_matprod_mat4")
    assert same(oracle.csmat_mul_csmat(m("mat1"), m("mat1_csc")), "mat1_self_matprod")
    r = oracle.csmat_mul_csmat(m("mat1_csc"), m("mat1"))      # CSC result ...
    assert r["storage"] == "CSC"
    ip, ix, dt = oracle.convert_storage(5, 5, r["indptr"], r["indices"], r["data"])  # ... .to_other_storage()
    e = m("mat1_self_matprod")
    assert np.array_equal(ip, e["indptr"]) and np.array_equal(ix, e["indices"]) and np.array_equal(dt, e["data"])


def test_mul_zero_rows(golden):
    # smmp.rs:475-489 (issue 239)
    fx = golden["mul_zero_rows"]
    z = np.zeros(0, dtype=np.uint64)
    shape, ip, ix, dt = oracle.mul_csr_csr(
        tuple(fx["a_shape"]), np.array(fx["a_indptr"], dtype=np.uint64), z, np.zeros(0),
        tuple(fx["b_shape"]), np.array(fx["b_indptr"], dtype=np.uint64), z, np.zeros(0))
    assert shape == tuple(fx["c_shape"]) and ix.size == fx["c_nnz"] and list(ip) == [0]


def test_one_long_row_multithreaded():
    # smmp.rs:503-513: 1 x 100 empty row times 100 x 10 zero matrix, Fixed(4)
    z = np.zeros(0, dtype=np.uint64)
    shape, ip, ix, dt, used = oracle.mul_csr_csr(
        (1, 100), np.zeros(2, dtype=np.uint64), z, np.zeros(0),
        (100, 10), np.zeros(101, dtype=np.uint64), z, np.zeros(0), threads=4, return_threads=True)
    assert shape == (1, 10) and ix.size == 0 and used == 1     # min(rows.max(1), 4)


def test_structure_only_fixtures(golden):
    # smmp.rs:515-555 (complex values; structure part) and block_matrix.rs:71-108
    fx = golden["mul_complex_structure"]
    ip, ix = np.array(fx["indptr"], dtype=np.uint64), np.array(fx["indices"], dtype=np.uint64)
    c_ip, c_ix = oracle.symbolic((4, 4), ip, ix, (4, 4), ip, ix)
    assert list(c_ip) == fx["c_indptr"] and list(c_ix) == fx["c_indices"]
    fx = golden["block_matrix_structure"]
    u = lambda k: np.array(fx[k], dtype=np.uint64)
    c_ip, c_ix = oracle.symbolic(tuple(fx["a_shape"]), u("a_indptr"), u("a_indices"),
                                 tuple(fx["b_shape"]), u("b_indptr"), u("b_indices"))
    assert list(c_ip) == fx["c_indptr"] and list(c_ix) == fx["c_indices"]


def test_structural_zeros_kept():
    # [1,-1] . [1;1] keeps one stored 0.0 (no numeric pruning, smmp.rs:109-119)
    u = lambda *v: np.array(v, dtype=np.uint64)
    shape, ip, ix, dt = oracle.mul_csr_csr((1, 2), u(0, 2), u(0, 1), np.array([1.0, -1.0]),
                                           (2, 1), u(0, 1, 2), u(0, 0), np.array([1.0, 1.0]))
    assert shape == (1, 1) and list(ip) == [0, 1] and list(ix) == [0] and dt[0] == 0.0


def test_dense_products_via_spmv(golden):
    # prod.rs:502-542, 580-595: csr_mulacc_dense_{row,col}maj == SpMV per rhs column
    for mk, dk, ek in (("mat1", "mat_dense1", "mat1_times_mat_dense1"),
                       ("mat5", "mat_dense2", "mat5_times_mat_dense2")):
        shape, ip, ix, dt = as_csr(golden[mk])
        rhs = np.array(golden[dk])
        exp = np.array(golden[ek]["rows"])
        out = np.zeros((shape[0], rhs.shape[1]))
        for j in range(rhs.shape[1]):
            col = np.zeros(shape[0])
            oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, rhs[:, j].copy(), col)
            out[:, j] = col
        assert np.all(np.abs(out - exp) <= golden[ek]["epsilon"])


def test_dot_product_row_primitive(golden):
    # vec.rs:1648-1673: sparse row . dense == 16 exactly (a 1-row SpMV)
    fx = golden["dot_product"]
    y = np.zeros(1)
    oracle.mul_acc_mat_vec_csr((1, fx["dim"]), np.array([0, 4], dtype=np.uint64),
                               np.array(fx["indices"], dtype=np.uint64), np.array(fx["data"]),
                               np.array(fx["dense"]), y)
    assert y[0] == fx["expected"]


def test_eye_identity():
    # csmat.rs:406-426 / lib.rs:52-73 doc tests: eye * x == x, eye * A == A  (config 1)
    shape, ip, ix, dt = oracle.eye(1000)
    x = 0.5 + np.arange(1000) / 1000.0
    y = np.zeros(1000)
    oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, x, y)
    assert np.array_equal(x, y)
    r = oracle.mul_csr_csr(shape, ip, ix, dt, shape, ip, ix, dt)
    assert np.array_equal(r[1], ip) and np.array_equal(r[2], ix) and np.array_equal(r[3], dt)


def test_gh374_index_overflow():
    # sprs/tests/gh374.rs:10-33: transposing needs row indices wider than I.
    # u32 twin of the u16 case: 2^32+... rows cannot be allocated here, so pin
    # the check itself: mat.rows() = 2^32 does not fit I = u32.
    ip = np.zeros(2, dtype=np.uint64)
    with pytest.raises(oracle.OracleError) as e:
        oracle.convert_storage(1, 1, ip, np.zeros(0, dtype=np.uint32), np.zeros(0), mat_rows=1 << 32)
    assert e.value.code == oracle.INDEX_OVERFLOW and "Index type is not large enough to hold" in str(e.value)


def test_grid_laplacian_layout():
    # examples/heat.rs:45-80
    shape, ip, ix, dt = oracle.grid_laplacian(4, 4)
    assert shape == (16, 16) and ix.size == 12 + 5 * 4
    r = 5                                   # interior vertex (1,1)
    s, e = int(ip[r]), int(ip[r + 1])
    assert list(ix[s:e]) == [1, 4, 5, 6, 9] and list(dt[s:e]) == [1, 1, -4, 1, 1]
    assert int(ip[1] - ip[0]) == 1 and ix[0] == 0 and dt[0] == 1.0
    oracle.check_structure(16, 16, ip, ix)


def test_sliced_non_proper_indptr(golden):
    # slice_outer (slicing.rs:65-89): indptr not rebased, indices/data offset
    shape, ip, ix, dt = as_csr(golden["mat1"])
    x = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    full = np.zeros(5)
    oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, x, full)
    part = np.zeros(3)
    s = int(ip[1])
    oracle.mul_acc_mat_vec_csr((3, 5), ip[1:5], ix[s:], dt[s:], x, part)
    assert np.array_equal(part, full[1:4])


def test_random_vs_scipy_cancellation_free():
    # independent cross-check: on positive data scipy's structure
    # equals sprs' after sort_indices()
    import scipy.sparse as sp
    a = sp.random(300, 200, density=0.05, random_state=3, format="csr")
    b = sp.random(200, 250, density=0.05, random_state=4, format="csr")
    a.data[:] = np.abs(a.data) + 0.5
    b.data[:] = np.abs(b.data) + 0.5
    a.sort_indices(); b.sort_indices()
    c = (a @ b).tocsr(); c.sort_indices()
    u = lambda v: v.astype(np.uint64)
    for t in (1, 3):
        shape, ip, ix, dt = oracle.mul_csr_csr((300, 200), u(a.indptr), u(a.indices), a.data,
                                               (200, 250), u(b.indptr), u(b.indices), b.data, threads=t)
        assert np.array_equal(ip, c.indptr) and np.array_equal(ix, c.indices)
        assert np.allclose(dt, c.data, rtol=1e-13, atol=0)


def test_bicgstab_reference_example(golden):
    """sprs/src/sparse/linalg/bicgstab.rs:336-369 (and the doc example :32-66): the solve must return Ok
    within 50 iterations at tol 1e-60 and A x must equal b to 1e-60 relative — i.e. the restatement has to
    reach a residual of exactly zero, which pins the order of every addition."""
    from oracle import oracle
    fx = golden["bicgstab_example"]
    ip, ix = np.array(fx["indptr"], dtype=np.uint64), np.array(fx["indices"], dtype=np.uint64)
    dt = np.array(fx["data"])
    x, info = oracle.bicgstab(tuple(fx["shape"]), ip, ix, dt, np.ones(4), np.ones(4), fx["tol"], fx["max_iter"],
                              storage="CSC")
    assert info["converged"] == 1 and info["iteration_count"] <= fx["max_iter"] and info["err"] < fx["tol"]
    dense = np.zeros((4, 4))
    for j in range(4):
        for p in range(int(ip[j]), int(ip[j + 1])):
            dense[int(ix[p]), j] = dt[p]
    assert np.all(np.abs(1.0 - np.ones(4) / (dense @ x)) < fx["tol"])
    # independent cross-check of the solution itself
    assert np.allclose(x, np.linalg.solve(dense, np.ones(4)), rtol=1e-14, atol=0)
    # u32 instantiation gives the same bits
    x32, info32 = oracle.bicgstab(tuple(fx["shape"]), ip.astype(np.uint32), ix.astype(np.uint32), dt, np.ones(4),
                                  np.ones(4), fx["tol"], fx["max_iter"], storage="CSC")
    assert np.array_equal(x, x32) and info == info32


def test_bicgstab_iteration_limit_is_not_an_error():
    """solve() returns Err(solver) with the last iterate when max_iter runs out (bicgstab.rs:168-170)"""
    import scipy.sparse as sp
    from oracle import oracle
    rng = np.random.default_rng(0)
    n = 300
    a = sp.random(n, n, density=0.02, random_state=1, format="csr")
    a = (a + sp.diags(np.abs(a).sum(axis=1).A1 + 1.0)).tocsr()
    a.sort_indices()
    b = rng.standard_normal(n)
    u = lambda v: v.astype(np.uint64)
    x, info = oracle.bicgstab((n, n), u(a.indptr), u(a.indices), a.data, np.zeros(n), b, 1e-300, 3)
    assert info["converged"] == 0 and info["iteration_count"] == 3
    x2, info2 = oracle.bicgstab((n, n), u(a.indptr), u(a.indices), a.data, np.zeros(n), b, 1e-11, 100)
    assert info2["converged"] == 1 and np.linalg.norm(a @ x2 - b) < 1e-11 and info2["hard_restart_count"] >= 1


def _py_gauss_seidel(n, ip, ix, dt, x, rhs, max_iter, eps):
    """heat.rs:103-139 line by line in Python (ndarray's sum = numeric_util::unrolled_fold, eight running sums)"""
    x = np.array(x, dtype=np.float64)

    def error():
        r = np.zeros(n)
        for i in range(n):
            s = 0.0
            for p in range(int(ip[i]), int(ip[i + 1])):
                s = s + dt[p] * x[int(ix[p])]
            r[i] = s - rhs[i]
        ps, k = [0.0] * 8, 0
        while n - k >= 8:
            for j in range(8):
                ps[j] = ps[j] + r[k + j]
            k += 8
        acc = 0.0
        for j in range(4):
            acc = acc + (ps[j] + ps[j + 4])
        for j in range(k, n):
            acc = acc + r[j]
        return float(np.sqrt(acc)) if acc >= 0 else float("nan")

    e = error()
    for it in range(max_iter):
        for row in range(n):
            sigma, diag = 0.0, None
            for p in range(int(ip[row]), int(ip[row + 1])):
                c = int(ix[p])
                if c != row:
                    sigma += dt[p] * x[c]
                else:
                    diag = dt[p]
            x[row] = (rhs[row] - sigma) / diag
        e = error()
        if e < eps:
            return x, (it, e, 1)
    return x, (max_iter, e, 0)


def test_gauss_seidel_heat_example():
    """sprs/examples/heat.rs:141-174 (the example's main: 10 x 10 grid, rhs = row + col on the border, 300 sweeps, eps 1e-8).
    The example asserts nothing and no Rust toolchain is here, so the C restatement is pinned three ways: it converges
    (the example prints "Solved system in N iterations"), its solution is the solution of the linear system, and an
    independent Python restatement written from the same source gives the same bits, sweep count and error."""
    from oracle import oracle
    shape, ip, ix, dt = oracle.grid_laplacian(10, 10)
    i, j = np.meshgrid(np.arange(10), np.arange(10), indexing="ij")
    border = (i == 0) | (i == 9) | (j == 0) | (j == 9)
    rhs = np.where(border, (i + j).astype(np.float64), 0.0).reshape(-1)
    x, info = oracle.gauss_seidel(shape, ip, ix, dt, np.zeros(100), rhs, 300, 1e-8)
    assert info["converged"] == 1 and info["iterations"] < 300 and info["error"] < 1e-8
    dense = np.zeros(shape)
    for r in range(100):
        for p in range(int(ip[r]), int(ip[r + 1])):
            dense[r, int(ix[p])] = dt[p]
    assert np.abs(x - np.linalg.solve(dense, rhs)).max() < 1e-12
    assert np.allclose(x.reshape(10, 10), i + j, rtol=0, atol=1e-12)       # harmonic boundary data: the solution is row + col
    x_py, (it, e, ok) = _py_gauss_seidel(100, ip, ix, dt, np.zeros(100), rhs, 300, 1e-8)
    assert np.array_equal(x, x_py) and (info["iterations"], info["converged"]) == (it, ok) and info["error"] == e
    # u32 instantiations: same bits
    for idx, ptr in ((np.uint32, np.uint32), (np.uint32, np.uint64)):
        x32, info32 = oracle.gauss_seidel(shape, ip.astype(ptr), ix.astype(idx), dt, np.zeros(100), rhs, 300, 1e-8)
        assert np.array_equal(x, x32) and info == info32


def test_gauss_seidel_iterates_and_edge_cases():
    """non-symmetric system against the Python restatement sweep by sweep (Err(error) after k sweeps, NaN error of a
    negative residual sum included); max_iter = 0 returns the start error; a row without diagonal panics (heat.rs:127)"""
    import scipy.sparse as sp
    from oracle import oracle
    rng = np.random.default_rng(5)
    n = 37
    a = sp.random(n, n, density=0.15, random_state=2, format="csr")
    a = (a + sp.diags(np.abs(a).sum(axis=1).A1 + 1.0)).tocsr()
    a.sort_indices()
    u = lambda v: v.astype(np.uint64)
    rhs, x0 = rng.standard_normal(n), rng.standard_normal(n)
    for k in (0, 1, 2, 5):
        x, info = oracle.gauss_seidel((n, n), u(a.indptr), u(a.indices), a.data, x0, rhs, k, -1.0)
        x_py, (it, e, ok) = _py_gauss_seidel(n, a.indptr, a.indices, a.data, x0, rhs, k, -1.0)
        assert np.array_equal(x, x_py) and info["iterations"] == it == k and info["converged"] == ok == 0
        assert (np.isnan(e) and np.isnan(info["error"])) or e == info["error"]
    with pytest.raises(oracle.OracleError):
        oracle.gauss_seidel((3, 3), np.array([0, 1, 2, 3], dtype=np.uint64), np.array([0, 0, 2], dtype=np.uint64), np.ones(3),
                            np.zeros(3), np.zeros(3), 1, 1e-8)


def test_triplets_to_cs_golden(golden):
    """sprs/src/sparse/triplet.rs:343-646 (triplet_incremental, _unordered, _additions, _from_vecs, _mutate_entry,
    _to_csr, _complex, _empty_lines): the oracle's restatement of TriMatIter::into_cs (triplet_iter.rs:127-224) must give
    the CSC the reference asserts, and for to_csr the reference's `expected.to_csr()` (its conversion, pinned above)."""
    cases = golden["triplet_cases"]
    assert {c["name"].split("#")[0] for c in cases} >= {"triplet_incremental", "triplet_unordered", "triplet_additions",
                                                         "triplet_from_vecs", "triplet_mutate_entry", "triplet_to_csr",
                                                         "triplet_complex", "triplet_empty_lines"}
    for c in cases:
        rows, cols = c["shape"]
        exp = c["csc"]
        ip, ix, dt = oracle.triplets_to_cs((rows, cols), c["rows"], c["cols"], c["data"], storage="CSC")
        assert ip.tolist() == exp["indptr"] and ix.tolist() == exp["indices"] and dt.tolist() == exp["data"], c["name"]
        # expected.to_csr(): convert the asserted CSC with the (pinned) storage conversion
        eip, eix, edt = oracle.convert_storage(cols, rows, np.array(exp["indptr"], dtype=np.uint64),
                                               np.array(exp["indices"], dtype=np.uint64), np.array(exp["data"]))
        rip, rix, rdt = oracle.triplets_to_cs((rows, cols), c["rows"], c["cols"], c["data"], storage="CSR")
        assert rip.tolist() == eip.tolist() and rix.tolist() == eix.tolist() and rdt.tolist() == edt.tolist(), c["name"]
