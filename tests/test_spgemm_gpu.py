"""GPU parity tests of the HIP SpGEMM path (smmp::mul_csr_csr twin) against the
CPU oracle and the reference's golden matrices.  Bar: indptr / indices
bit-exact, values within 1e-10 relative (north star)."""
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


def gpu_mul(a, b, validate=True):
    from sprs_amd import smmp
    from sprs_amd.device import DeviceCsMat
    da = DeviceCsMat.from_host(*a, validate=validate)
    db = DeviceCsMat.from_host(*b, validate=validate)
    c = smmp.mul_csr_csr(da, db)
    assert c.is_csr()
    return c.to_host()


def check_against_oracle(a, b, exact_values=False, ref=None):
    from oracle import oracle
    shape, ip, ix, dt = gpu_mul(a, b)
    rshape, rip, rix, rdt = ref if ref is not None else oracle.mul_csr_csr(*a, *b, threads=1)
    assert shape == rshape
    assert ip.dtype == rip.dtype and ix.dtype == rix.dtype          # same I / Iptr as the operands (smmp.rs:196-199)
    assert np.array_equal(ip, rip), "indptr differs"
    assert np.array_equal(ix, rix), "indices differ"
    if exact_values:
        assert np.array_equal(dt, rdt)
    assert rel_err(dt, rdt) <= TOL
    return shape, ip, ix, dt


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_golden_mul_csr_csr(hip, golden, idx, ptr):
    # prod.rs:425-436, smmp.rs:467-473: exact equality with the fixtures
    a, b = as_csr(golden["mat1"], idx, ptr), as_csr(golden["mat2"], idx, ptr)
    for rhs, key in ((a, "mat1_self_matprod"), (b, "mat1_matprod_mat2")):
        exp = as_csr(golden[key], idx, ptr)
        shape, ip, ix, dt = gpu_mul(a, rhs)
        assert shape == exp[0]
        assert np.array_equal(ip, exp[1]) and np.array_equal(ix, exp[2]) and np.array_equal(dt, exp[3])


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_symbolic_and_numeric_twins(hip, golden, idx, ptr):
    """smmp.rs:422-465 `symbolic_and_numeric`: the two halves of the product as separate calls — the
    structure alone (values 0.0), then the values into that structure; re-use of the structure with other
    values of the same patterns; and what the reference does not check: a C of another structure is refused."""
    from oracle import oracle
    from sprs_amd import _ffi, gen, smmp
    from sprs_amd.device import DeviceCsMat
    a, b, exp = (as_csr(golden[k], idx, ptr) for k in ("mat1", "mat2", "mat1_matprod_mat2"))
    da, db = DeviceCsMat.from_host(*a), DeviceCsMat.from_host(*b)
    c = smmp.symbolic(da, db)
    shape, ip, ix, dt = c.to_host()
    assert shape == exp[0] and np.array_equal(ip, exp[1]) and np.array_equal(ix, exp[2])
    assert ip.dtype == exp[1].dtype and ix.dtype == exp[2].dtype and not dt.any()
    smmp.numeric(da, db, c)
    assert np.array_equal(c.to_host()[3], exp[3])
    # same patterns, other values: numeric alone refreshes C (the point of the split)
    da2 = DeviceCsMat.from_host(a[0], a[1], a[2], a[3] * 3.0)
    smmp.numeric(da2, db, c)
    assert np.array_equal(c.to_host()[3], exp[3] * 3.0)
    # a large-row case against the oracle's two halves
    n = 12000
    indptr, indices, data = gen.rmat_csr(n, 8, seed=13)
    m = ((n, n), indptr.numpy().astype(ptr), indices.numpy().astype(idx), data.numpy())
    dm = DeviceCsMat.from_host(*m)
    cs = smmp.symbolic(dm, dm)
    r_ip, r_ix = oracle.symbolic(m[0], m[1], m[2], m[0], m[1], m[2])
    got = cs.to_host()
    assert np.array_equal(got[1], r_ip) and np.array_equal(got[2], r_ix)
    smmp.numeric(dm, dm, cs)
    assert np.array_equal(cs.to_host()[3], oracle.numeric(*m, *m, r_ip, r_ix))
    assert np.diff(r_ip.astype(np.int64)).max() > 4096
    # wrong structure: another matrix of the right shape
    with pytest.raises(_ffi.SprsHipError) as e:
        smmp.numeric(dm, dm, DeviceCsMat.from_host(*m))
    assert e.value.status == _ffi.BAD_STRUCTURE
    with pytest.raises(_ffi.SprsHipError) as e:
        smmp.numeric(da, db, cs)                                   # wrong shape
    assert e.value.status == _ffi.DIM_MISMATCH


def test_golden_structure_only(hip, golden):
    # smmp.rs:515-555 (complex; structure) and tests/block_matrix.rs:71-108
    fx = golden["mul_complex_structure"]
    m = (tuple(fx["shape"]), np.array(fx["indptr"], dtype=np.uint64), np.array(fx["indices"], dtype=np.uint64),
         np.ones(len(fx["indices"])))
    shape, ip, ix, dt = gpu_mul(m, m)
    assert list(ip) == fx["c_indptr"] and list(ix) == fx["c_indices"]
    fx = golden["block_matrix_structure"]
    u = lambda k: np.array(fx[k], dtype=np.uint64)
    a = (tuple(fx["a_shape"]), u("a_indptr"), u("a_indices"), np.ones(3))
    b = (tuple(fx["b_shape"]), u("b_indptr"), u("b_indices"), np.ones(2))
    shape, ip, ix, dt = gpu_mul(a, b)
    assert list(ip) == fx["c_indptr"] and list(ix) == fx["c_indices"]


def test_zero_rows_and_empty(hip, golden):
    # smmp.rs:475-489 (issue 239) and smmp.rs:503-513
    fx = golden["mul_zero_rows"]
    z = np.zeros(0, dtype=np.uint64)
    a = (tuple(fx["a_shape"]), np.array(fx["a_indptr"], dtype=np.uint64), z, np.zeros(0))
    b = (tuple(fx["b_shape"]), np.array(fx["b_indptr"], dtype=np.uint64), z, np.zeros(0))
    shape, ip, ix, dt = gpu_mul(a, b)
    assert shape == tuple(fx["c_shape"]) and ix.size == 0 and list(ip) == [0]
    a = ((1, 100), np.zeros(2, dtype=np.uint64), z, np.zeros(0))
    b = ((100, 10), np.zeros(101, dtype=np.uint64), z, np.zeros(0))
    shape, ip, ix, dt = gpu_mul(a, b)
    assert shape == (1, 10) and ix.size == 0 and list(ip) == [0, 0]


def test_structural_zeros_kept(hip):
    # products that cancel or are zero stay stored (smmp.rs:109-119)
    u = lambda *v: np.array(v, dtype=np.uint64)
    a = ((1, 2), u(0, 2), u(0, 1), np.array([1.0, -1.0]))
    b = ((2, 1), u(0, 1, 2), u(0, 0), np.array([1.0, 1.0]))
    shape, ip, ix, dt = gpu_mul(a, b)
    assert shape == (1, 1) and list(ip) == [0, 1] and list(ix) == [0] and dt[0] == 0.0
    a = ((1, 1), u(0, 1), u(0), np.array([0.0]))
    b = ((1, 3), u(0, 2), u(0, 2), np.array([5.0, np.inf]))
    shape, ip, ix, dt = gpu_mul(a, b)
    assert list(ix) == [0, 2] and dt[0] == 0.0 and np.isnan(dt[1])


def test_eye_times_matrix(hip, golden):
    # lib.rs:52-73 doc test: eye * A == A
    from oracle import oracle
    a = as_csr(golden["mat5"])                       # 5 x 15
    e = oracle.eye(5)
    shape, ip, ix, dt = gpu_mul(e, a)
    assert shape == (5, 15) and np.array_equal(ip, a[1]) and np.array_equal(ix, a[2]) and np.array_equal(dt, a[3])


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_rmat_square_vs_oracle(hip, idx, ptr):
    # config 5 shape at test size: A*A, power-law rows; small + large row paths
    from sprs_amd import gen
    n = 20000
    indptr, indices, data = gen.rmat_csr(n, 8, seed=5)
    a = ((n, n), indptr.numpy().astype(ptr), indices.numpy().astype(idx), data.numpy())
    shape, ip, ix, dt = check_against_oracle(a, a, exact_values=True)
    lens = np.diff(ip.astype(np.int64))
    assert lens.max() > 4096 and (lens == 0).any()


def test_rectangular_random_mixed_sign(hip):
    import scipy.sparse as sp
    a = sp.random(700, 3000, density=0.01, random_state=1, format="csr")
    b = sp.random(3000, 900, density=0.02, random_state=2, format="csr")
    a.data[:] = np.random.default_rng(0).standard_normal(a.nnz)
    b.data[:] = np.random.default_rng(1).standard_normal(b.nnz)
    a.sort_indices(); b.sort_indices()
    u = lambda v: v.astype(np.uint64)
    A = ((700, 3000), u(a.indptr), u(a.indices), a.data)
    B = ((3000, 900), u(b.indptr), u(b.indices), b.data)
    check_against_oracle(A, B, exact_values=True)     # same k order, unfused: bit-identical values


def test_multi_window_columns(hip):
    # B wider than one 2^19-column window: rows above the small-row threshold span 3 windows
    cols = (1 << 20) + 12345
    rng = np.random.default_rng(3)
    b_lens = [int(v) for v in rng.integers(200, 900, size=40)]
    B = ragged_csr(b_lens, cols, seed=9)
    a_lens = [40, 0, 3, 25, 1, 40]
    A = ragged_csr(a_lens, 40, seed=4)
    shape, ip, ix, dt = check_against_oracle(A, B, exact_values=True)
    assert ix.max() >= (1 << 20)
    # columns exactly at the window edges
    u = lambda *v: np.array(v, dtype=np.uint64)
    edge = np.array(sorted(set([0, (1 << 19) - 1, 1 << 19, (1 << 20) - 1, 1 << 20, cols - 1] +
                               list(range(1000, 1700)))), dtype=np.uint64)
    B2 = ((1, cols), u(0, edge.size), edge, np.full(edge.size, 2.0))
    A2 = ((2, 1), u(0, 1, 2), u(0, 0), np.array([3.0, 0.5]))
    check_against_oracle(A2, B2, exact_values=True)


def _order_stress_cases():
    """Inputs on which ANY reordering of the additions into one C(i,j) changes the bits: many k's per row whose B rows
    are short and crowd onto few columns (every wave instruction of the value walk holds many k-runs, every accumulator a
    long chain), magnitudes spread over 12 decades, mixed signs.  The first four shapes take the four routes of the large-row
    kernel, the last three those of the wave-per-row kernel: one staged group with the window kept in registers (<= 2048 products), one group with a second walk,
    several staged groups (> 256 k's), several passes (> 3072 outputs in a window)."""
    cases = []
    rng = np.random.default_rng(77)
    u = lambda v: np.asarray(v, dtype=np.uint64)
    def crowd(n_k, per_row, cols, hot, stride=1):
        # B: n_k rows of `per_row` distinct columns, most of them among `hot` popular ones (the first, or every stride-th)
        ip = [0]; ix = []; dt = []
        for _ in range(n_k):
            n_hot = min(hot, max(1, int(per_row * 0.8)))
            c = set((rng.choice(hot, size=n_hot, replace=False) * stride).tolist())
            while len(c) < per_row:
                c.add(int(rng.integers(0, cols)))
            c = sorted(c)
            ix += c
            dt += list(rng.standard_normal(len(c)) * 10.0 ** rng.integers(-6, 7, size=len(c)))
            ip.append(len(ix))
        return (n_k, cols), u(ip), u(ix), np.asarray(dt)
    def rows_over(n_rows, n_k, density):
        m = rng.random((n_rows, n_k)) < density
        m[:, 0] = True
        ip = np.concatenate([[0], np.cumsum(m.sum(1))])
        ix = np.concatenate([np.nonzero(r)[0] for r in m])
        dt = rng.standard_normal(ix.size) * 10.0 ** rng.integers(-3, 4, size=ix.size)
        return (n_rows, n_k), u(ip), u(ix), dt
    cases.append((rows_over(6, 200, 0.9), crowd(200, 8, 5000, 24)))         # ~1 400 products a row: kept in registers
    cases.append((rows_over(6, 250, 0.95), crowd(250, 30, 5000, 60)))       # ~7 000 products: second walk, one pass
    cases.append((rows_over(4, 700, 0.9), crowd(700, 6, 200_000, 16)))      # three staged groups, two windows
    cases.append((rows_over(3, 240, 0.95), crowd(240, 900, 40_000, 4000)))  # ~4 000+ outputs in the window: passes
    # rows of <= 64 k's: the wave-per-row kernel — windows kept in registers (<= 256 entries per 8192 columns), windows
    # walked twice, and windows of more than 512 outputs (several passes over the window's entries)
    cases.append((rows_over(8, 60, 0.9), crowd(60, 14, 100_000, 300, stride=333)))
    cases.append((rows_over(8, 64, 0.95), crowd(64, 40, 5000, 50)))
    cases.append((rows_over(4, 60, 0.95), crowd(60, 900, 40_000, 4000)))
    return cases


def test_order_of_additions_stress(hip):
    for a, b in _order_stress_cases():
        for lane_order, atomic in ((0, 1), (2, 1), (2, 0)):       # one ds_add_f64 per wave instruction / one per k-run / read-add-write per k-run
            hip.set_option("spgemm_lane_order", lane_order)
            hip.set_option("spgemm_lds_atomic", atomic)
            try:
                check_against_oracle(a, b, exact_values=True)
            finally:
                hip.set_option("spgemm_lane_order", 0)
                hip.set_option("spgemm_lds_atomic", 1)


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_row_class_boundaries(hip, idx, ptr):
    """Rows sitting exactly on the thresholds that route them (row_work_kernel): 64 / 65 and 512 / 513 products (hash
    tables), 64 / 65 k's and `spgemm_mid` products (wave-per-row vs workgroup kernel), `spgemm_heavy` (one task per
    window), with B wider than one 2^17-column window and columns on the window / bucket edges, empty rows of B among the
    k's, for every index type."""
    rng = np.random.default_rng(5)
    cols = (1 << 17) * 2 + 4321
    n_b = 200
    b_lens = [0, 1, 2, 63, 64, 65, 511, 512, 513] + [int(v) for v in rng.integers(0, 700, size=n_b - 9)]
    b_lens[20] = 0
    B = ragged_csr(b_lens, cols, seed=21, idx=idx, ptr=ptr, positive=False)
    # put entries exactly on window / bucket edges into a few rows of B
    shape, bip, bix, bdt = B
    edges = np.array([0, 2047, 2048, (1 << 14) - 1, 1 << 14, (1 << 16), (1 << 17) - 1, 1 << 17, (1 << 18) - 1, 1 << 18, cols - 1], dtype=np.int64)
    for r in (30, 31, 32):
        s0, e0 = int(bip[r]), int(bip[r + 1])
        if e0 - s0 >= edges.size:
            seg = np.array(sorted(set(edges.tolist() + bix[s0:e0].astype(np.int64).tolist())))[: e0 - s0]
            # keep the row length: take the edges and fill with the smallest other columns
            seg = np.array(sorted(set(edges.tolist()) | set(seg.tolist())))[: e0 - s0] if seg.size >= e0 - s0 else seg
            if seg.size == e0 - s0:
                bix[s0:e0] = np.sort(seg).astype(idx)
    B = (shape, bip, bix, bdt)
    lens = np.diff(bip.astype(np.int64))

    def row_with(products_at_least, n_k):
        ks = np.sort(rng.choice(n_b, size=n_k, replace=False))
        return ks

    a_rows = []
    # k counts around 64, products small and large
    for n_k in (1, 2, 63, 64, 65, 100, 199):
        a_rows.append(row_with(0, n_k))
    # single k's of exactly 64 / 65 / 512 / 513 entries: the hash-table thresholds
    for k in (4, 5, 7, 8, 20):
        a_rows.append(np.array([k]))
    a_rows.append(np.array([], dtype=np.int64))
    a_ip = np.zeros(len(a_rows) + 1, dtype=np.int64)
    a_ip[1:] = np.cumsum([r.size for r in a_rows])
    a_ix = np.concatenate(a_rows) if a_rows else np.zeros(0, dtype=np.int64)
    a_dt = rng.standard_normal(a_ix.size) * 10.0 ** rng.integers(-4, 5, size=a_ix.size)
    A = ((len(a_rows), n_b), a_ip.astype(ptr), a_ix.astype(idx), a_dt)
    ub = [int(lens[r].sum()) for r in a_rows]
    assert 64 in ub and 65 in ub and 512 in ub and 513 in ub
    for mid, heavy in ((65536, 131072), (max(ub[3] - 1, 513), 131072), (ub[3], 1024), (0, 2048)):
        hip.set_option("spgemm_mid", mid)
        hip.set_option("spgemm_heavy", heavy)
        try:
            check_against_oracle(A, B, exact_values=True)
        finally:
            hip.set_option("spgemm_mid", 65536)
            hip.set_option("spgemm_heavy", 131072)


def test_hub_rows_of_the_work_pass(hip):
    """row_work_kernel walks a row of A with 16 lanes; one stride for rows of at most 16 k's, four unconditional strides beyond, and
    the whole wave for a hub row (>= 2048 k's): rows of 0, 1, 16, 17, 64, 65, 2047, 2048, 2049 and 5000 k's next to each other (the
    lengths on both sides of every switch, hub rows in different lane groups of one wave and alone in theirs), B with short rows —
    products per row, classes, the extents the micro rows read and the whole product against the oracle, bits included."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    inner, cols = 6000, 900
    b_lens = rng.integers(0, 4, size=inner)
    b_ip = np.zeros(inner + 1, dtype=np.int64)
    b_ip[1:] = np.cumsum(b_lens)
    b_ix = np.concatenate([np.sort(rng.choice(cols, size=int(l), replace=False)) for l in b_lens]).astype(np.int64)
    b_dt = rng.standard_normal(b_ix.size)
    a_lens = [0, 1, 16, 17, 2048, 64, 65, 2047, 2049, 3, 5000, 2, 2048, 2048, 2048, 2048, 5, 0]
    a_ip = np.zeros(len(a_lens) + 1, dtype=np.int64)
    a_ip[1:] = np.cumsum(a_lens)
    a_ix = np.concatenate([np.sort(rng.choice(inner, size=l, replace=False)) for l in a_lens]).astype(np.int64)
    a_dt = rng.standard_normal(a_ix.size)
    a = ((len(a_lens), inner), a_ip.astype(np.uint64), a_ix.astype(np.uint64), a_dt)
    b = ((inner, cols), b_ip.astype(np.uint64), b_ix.astype(np.uint64), b_dt)
    ref = oracle.mul_csr_csr(*a, *b, threads=1)
    _, ip, ix, dt = gpu_mul(a, b)
    assert np.array_equal(ip, ref[1]) and np.array_equal(ix, ref[2])
    assert np.array_equal(dt.view(np.uint64), ref[3].view(np.uint64))


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_micro_rows_lane_groups(hip, idx, ptr):
    """Rows of at most 16 / 32 / 64 products AND k's run on lane groups (micro_rows_kernel: 4 / 2 / 1 rows per wave, no LDS):
    rows sitting exactly on 16 / 17, 32 / 33, 64 / 65 products; rows whose k's outnumber their products (k's with empty rows of
    B: 17 and 39 k's for 12 and 22 products — must fall to the wider group or to the hash kernel); duplicate columns across k's (the
    first-occurrence / ordered-sum logic: values of wildly different magnitude, bits compared); a last group with fewer rows
    than the wave has room for; the reference's own benchmark shape in small (uniform density 4 / cols, sprs-benches
    main.rs:148-163).  Structure and value BITS against the oracle, with the hash kernel (spgemm_micro = 2) as the A/B."""
    from oracle import oracle
    from sprs_amd import gen
    rng = np.random.default_rng(77)
    n_b, cols = 120, 700
    b_lens = [0] * 30 + [1] * 10 + [2] * 10 + [3] * 10 + [4] * 20 + [8] * 10 + [16] * 10 + [17, 31, 32, 33, 63, 64, 65, 5, 6, 7] + [12] * 10
    B = ragged_csr(b_lens, cols, seed=31, idx=idx, ptr=ptr, positive=False)
    _, bip, bix, bdt = B
    bdt[:] = rng.standard_normal(bdt.size) * 10.0 ** rng.integers(-12, 13, size=bdt.size)      # order-sensitive sums
    lens = np.diff(bip.astype(np.int64))
    pick = lambda want: [int(k) for k in np.nonzero(lens == want)[0]]
    empty, ones, fours, eights, sixteens, twelves = pick(0), pick(1), pick(4), pick(8), pick(16), pick(12)
    a_rows = [
        fours[:4],                       # 16 products, 4 k's
        fours[:4] + ones[:1],            # 17 products
        eights[:4],                      # 32
        eights[:4] + ones[:1],           # 33
        sixteens[:4],                    # 64
        sixteens[:4] + ones[:1],         # 65: not a micro row any more
        empty[:16] + twelves[:1],        # 12 products, 17 k's: too many k's for a group of 16
        empty[:28] + twelves[:1] + ones[:10],   # 22 products, 39 k's: too many k's for a group of 32
        pick(17), pick(31), pick(32), pick(33), pick(63), pick(64), pick(65),   # single k's on the edges
        twelves[:5],                     # 60 products with many columns shared between the k's
        [],                              # empty row
        ones[:3],                        # a lone short row at the end of its class list
    ]
    a_rows = [sorted(r) for r in a_rows]
    a_ip = np.zeros(len(a_rows) + 1, dtype=np.int64)
    a_ip[1:] = np.cumsum([len(r) for r in a_rows])
    a_ix = np.array([k for r in a_rows for k in r], dtype=np.int64)
    a_dt = rng.standard_normal(a_ix.size) * 10.0 ** rng.integers(-12, 13, size=a_ix.size)
    A = ((len(a_rows), n_b), a_ip.astype(ptr), a_ix.astype(idx), a_dt)
    ub = [int(sum(lens[k] for k in r)) for r in a_rows]
    assert [16, 17, 32, 33, 64, 65, 12, 22] == ub[:8]
    cases = [(A, B)]
    for n, per, seed in ((3000, 4, 1), (500, 30, 4)):                   # uniform density: every row a micro row / most of them
        a = gen.uniform_csr((n, n), per / n, seed=seed, value_seed=seed + 50)
        b = gen.uniform_csr((n, n), per / n, seed=seed + 100, value_seed=seed + 150)
        cases.append(tuple(((n, n), m[0].numpy().astype(ptr), m[1].numpy().astype(idx), m[2].numpy()) for m in (a, b)))
    # long runs of one column: 60 k's whose rows of B all hold column 7 (and one or no other column) — a run as long as the group
    # is wide, added in k order from 0.0 (smmp.rs:166-181); and rows of B made of the same 3 columns
    nb2 = 64
    b2_rows = [[7] if k % 3 else [7, 100 + k] for k in range(60)] + [[1, 2, 3]] * 4
    b2_ip = np.zeros(nb2 + 1, dtype=np.int64)
    b2_ip[1:] = np.cumsum([len(r) for r in b2_rows])
    b2_ix = np.array([c for r in b2_rows for c in r], dtype=np.int64)
    b2_dt = rng.standard_normal(b2_ix.size) * 10.0 ** rng.integers(-12, 13, size=b2_ix.size)
    a2_rows = [list(range(60)), list(range(0, 40)), list(range(5, 21)), [60, 61, 62, 63], list(range(50, 64)), list(range(0, 60, 3))]
    a2_ip = np.zeros(len(a2_rows) + 1, dtype=np.int64)
    a2_ip[1:] = np.cumsum([len(r) for r in a2_rows])
    a2_ix = np.array([k for r in a2_rows for k in r], dtype=np.int64)
    a2_dt = rng.standard_normal(a2_ix.size) * 10.0 ** rng.integers(-12, 13, size=a2_ix.size)
    cases.append((((len(a2_rows), nb2), a2_ip.astype(ptr), a2_ix.astype(idx), a2_dt), ((nb2, 200), b2_ip.astype(ptr), b2_ix.astype(idx), b2_dt)))
    # the same rows against a B of 2^26 + 100 columns: past the width where a (column, position) sort key fits one 32-bit word
    wide = (1 << 26) + 100
    scale = wide // cols
    cases.append((A, ((n_b, wide), bip, (bix.astype(np.int64) * scale + 3).astype(idx), bdt)))
    for a, b in cases:
        ref = oracle.mul_csr_csr(*a, *b, threads=1)
        for micro in (0, 2):                                              # lane groups, the hash kernel
            hip.set_option("spgemm_micro", micro)
            try:
                _, ip, ix, dt = gpu_mul(a, b)
            finally:
                hip.set_option("spgemm_micro", 0)
            assert np.array_equal(ip, ref[1]) and np.array_equal(ix, ref[2])
            assert np.array_equal(dt.view(np.uint64), ref[3].view(np.uint64)), "value bits differ (micro = %d)" % micro


_DENSE_CASE = []


def _dense_output_case():
    """built once per session: scipy.sparse.random needs ~20 s for these shapes"""
    if not _DENSE_CASE:
        import scipy.sparse as sp
        rng = np.random.default_rng(16)
        cols = 300_000
        # (numpy + coo instead of scipy.sparse.random, which needs ~20 s for these shapes)
        per_row = 1200
        r = np.repeat(np.arange(2000), per_row)
        b = sp.coo_matrix((np.ones(r.size), (r, rng.integers(0, cols, r.size))), shape=(2000, cols)).tocsr()
        band = sp.csr_matrix((rng.random((2000, 6000)) < 0.5).astype(np.float64))       # dense outputs in cols < 6000
        b = (b + sp.hstack([band, sp.csr_matrix((2000, cols - 6000))])).tocsr()
        a = sp.csr_matrix((rng.random((24, 2000)) < 0.3).astype(np.float64))            # ~600 k's per row
        a.data[:] = rng.standard_normal(a.nnz)
        b.data[:] = rng.standard_normal(b.nnz)
        a.sort_indices(); b.sort_indices()
        u = lambda v: v.astype(np.uint64)
        from oracle import oracle
        A = ((24, 2000), u(a.indptr), u(a.indices), a.data)
        B = ((2000, cols), u(b.indptr), u(b.indices), b.data)
        _DENSE_CASE.append((A, B, oracle.mul_csr_csr(*A, *B, threads=1)))
    return _DENSE_CASE[0]


@pytest.mark.parametrize("winlog", [16, 17, 18, 19])
def test_window_sizes_and_dense_outputs(hip, winlog):
    """Every LDS layout of the large-row kernels (option spgemm_winlog), on inputs that stress the
    ordered accumulation: > 256 k's per row (several staged groups), windows whose outputs are
    nearly dense (every chunk of the expansion holds duplicates -> several ordering rounds, several
    passes), heavy rows narrowed to 2^13-column windows, mixed signs (any reordering of the
    additions would change the bits)."""
    import scipy.sparse as sp
    hip.set_option("spgemm_winlog", winlog)
    hip.set_option("spgemm_heavy", 4096)
    try:
        A, B, ref = _dense_output_case()
        for bucket in (1, 0):
            hip.set_option("spgemm_bucket", bucket)
            shape, ip, ix, dt = check_against_oracle(A, B, exact_values=True, ref=ref)
        assert np.diff(ip.astype(np.int64)).max() > 100_000
    finally:
        hip.set_option("spgemm_winlog", 17)
        hip.set_option("spgemm_heavy", 131072)
        hip.set_option("spgemm_bucket", 1)


def test_result_pool_reuse_and_trim(hip):
    """Released result blocks are reused by the next product (same bits out), are capped by
    pool_max_bytes, and go back to the driver on pool_trim / option pool = 0."""
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat
    n = 60000
    indptr, indices, data = gen.rmat_csr(n, 8, seed=11)
    a = DeviceCsMat.from_host((n, n), indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64),
                              data.numpy())
    hip.pool_trim()
    assert hip.get_option("pool") == 1 and hip.get_option("pool_cached_bytes") == 0
    c1 = a * a
    ref = c1.to_host()
    nbytes = c1.nnz() * 16
    assert nbytes > (2 << 20)
    del c1
    assert hip.get_option("pool_cached_bytes") >= nbytes            # both blocks came back
    c2 = a * a                                                       # ... and serve the next result
    assert hip.get_option("pool_cached_bytes") < nbytes              # (what is cached now: the product's own temporaries)
    for x, y in zip(ref[1:], c2.to_host()[1:]):
        assert np.array_equal(x, y)
    del c2
    assert hip.pool_trim() >= nbytes and hip.get_option("pool_cached_bytes") == 0
    hip.set_option("pool_max_bytes", 1 << 20)                        # nothing this big may be kept (small temporaries may)
    c3 = a * a
    del c3
    assert hip.get_option("pool_cached_bytes") <= 1 << 20
    hip.set_option("pool_max_bytes", 128 << 30)
    hip.set_option("pool", 0)
    c4 = a * a
    del c4
    assert hip.get_option("pool_cached_bytes") == 0
    hip.set_option("pool", 1)


def test_row_blocks_virtual_ranks(hip):
    """The multi-GPU SpGEMM driver (sprs_amd/dist.py RowShardedSpGEMM: A by product-balanced row blocks, B
    replicated, no exchange) with 4 virtual ranks on the one GPU: the blocks of C, one after the other,
    are the rows of the single-handle product bit for bit."""
    import torch
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat
    from sprs_amd.dist import RowShardedSpGEMM
    dev = torch.device("cuda", 0)
    n = 50000
    indptr, indices, data = gen.rmat_csr(n, 8, seed=31, device=dev)
    a = ((n, n), indptr, indices, data)
    whole = (DeviceCsMat.wrap_torch(*a) * DeviceCsMat.wrap_torch(*a)).to_host()
    cuts_seen, prods = None, []
    for g in range(4):
        sh = RowShardedSpGEMM(a, a, virtual=(g, 4))
        shape, ip, ix, dt = sh.multiply().to_host()
        lo, hi = int(whole[1][sh.r0]), int(whole[1][sh.r1])
        assert shape == (sh.r1 - sh.r0, n)
        assert np.array_equal(ip + np.uint64(lo), whole[1][sh.r0:sh.r1 + 1])
        assert np.array_equal(ix, whole[2][lo:hi]) and np.array_equal(dt, whole[3][lo:hi])
        assert cuts_seen in (None, sh.cuts)
        cuts_seen = sh.cuts
        prods.append(sh.block_products)
    assert cuts_seen[0] == 0 and cuts_seen[-1] == n and max(prods) <= sum(prods) / 4 * 1.3


def test_contract_violations(hip, golden):
    from sprs_amd import SprsHipError, _ffi, smmp
    from sprs_amd.device import DeviceCsMat
    a = DeviceCsMat.from_host(*as_csr(golden["mat3"]))              # 5 x 4
    with pytest.raises(SprsHipError, match="Dimension mismatch") as e:   # smmp.rs:207
        smmp.mul_csr_csr(a, a)
    assert e.value.status == _ffi.DIM_MISMATCH
    shape, ip, ix, dt = as_csr(golden["mat1_csc"])
    csc = DeviceCsMat.from_host(shape, ip, ix, dt, storage=_ffi.CSC)
    sq = DeviceCsMat.from_host(*as_csr(golden["mat1"]))
    with pytest.raises(SprsHipError) as e:
        smmp.mul_csr_csr(sq, csc)
    assert e.value.status == _ffi.STORAGE_MISMATCH
    b32 = DeviceCsMat.from_host(*as_csr(golden["mat1"], np.uint32, np.uint32))
    with pytest.raises(SprsHipError) as e:                         # operands share I / Iptr (smmp.rs:196-199)
        smmp.mul_csr_csr(sq, b32)
    assert e.value.status == _ffi.STORAGE_MISMATCH


def test_iptr_overflow_u32(hip):
    # Iptr::from_usize panics when nnz(C) does not fit (smmp.rs:121; indexing.rs:104-108):
    # 70000 x 1 times 1 x 70000 of ones has 4.9e9 > 2^32 stored entries.
    from sprs_amd import SprsHipError, _ffi, smmp
    from sprs_amd.device import DeviceCsMat
    n = 70000
    a = DeviceCsMat.from_host((n, 1), np.arange(n + 1, dtype=np.uint32), np.zeros(n, dtype=np.uint32), np.ones(n))
    b = DeviceCsMat.from_host((1, n), np.array([0, n], dtype=np.uint32), np.arange(n, dtype=np.uint32), np.ones(n))
    with pytest.raises(SprsHipError, match="Index type is not large enough") as e:
        smmp.mul_csr_csr(a, b)
    assert e.value.status == _ffi.INDEX_OVERFLOW


def test_deterministic_and_operator(hip):
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat
    n = 30000
    indptr, indices, data = gen.rmat_csr(n, 6, seed=21)
    a = DeviceCsMat.from_host((n, n), indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64),
                              data.numpy())
    r1 = (a * a).to_host()            # `&A * &A` -> csmat_mul_csmat -> smmp::mul_csr_csr (csmat.rs:1866-1949)
    r2 = (a * a).to_host()
    for x, y in zip(r1[1:], r2[1:]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("idx_bytes", [8, 4])
def test_config5_full_size_row_blocks(hip, idx_bytes):
    """BASELINE config 5 at FULL size: A*A for R-MAT 1M x 1M, ~8 nnz/row -> nnz(C) = 3.3e9 (53 GB on the
    device with usize indices).  The host oracle cannot hold C, so three row blocks are recomputed by the
    oracle (smmp on A[rows,:] x A) and compared entry by entry — structure AND values bit for bit, the
    additions happen in the reference's order; plus size-independent structure checks on the device:
    indptr non-decreasing and consistent with nnz, every row strictly increasing."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "spgemm_bench.py"), "1000000", "8", str(idx_bytes),
                        "300"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["nnz_c"] > 3_000_000_000 and d["nnz_a"] > 7_000_000
    assert d["parity"]["structure_bit_exact"] and d["parity"]["max_rel_err"] <= TOL
    assert d["parity"]["values_bit_exact"], d["parity"]
    assert d["structure_checks"]["rows_strictly_increasing"] and d["structure_checks"]["indptr_monotone"]


def test_config5_developer_build(hip):
    """The hang guard: config 5 once on the DEVELOPER library (make DEVTOOLS=1, libsprs_hip_dev.so), under a timeout.  In
    round 3 a mis-compiled row loop (`for (;;)` + `break` around a wave-uniform draw) made config 5 spin forever in
    one build variant only while small products and the CPU emulator passed: both libraries therefore run the full-size
    product in this suite, and a variant that no longer terminates fails here instead of hanging a bench."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev_lib = os.path.join(root, "sprs_amd", "libsprs_hip_dev.so")
    if not os.path.exists(dev_lib):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(root, "sprs_amd", "csrc"), "DEVTOOLS=1"])
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "spgemm_bench.py"), "1000000", "8", "8", "100"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SPRS_HIP_LIBRARY=dev_lib))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["nnz_c"] > 3_000_000_000
    assert d["parity"]["structure_bit_exact"] and d["parity"]["values_bit_exact"], d["parity"]


def test_config5_five_products_bit_identical(hip):
    """BASELINE config 5 at full size, five times: nnz, indptr and position-weighted 64-bit checksums of the
    indices and of the value BITS must be identical from product to product (the accumulation order is fixed:
    no float atomics; the ordering race noted in spgemm.hip would show up here)."""
    import ctypes as C
    import torch
    from sprs_amd import _ffi, gen, smmp
    from sprs_amd.device import DeviceCsMat
    dev = torch.device("cuda", 0)
    n = 1_000_000
    indptr, indices, data = gen.rmat_csr(n, 8, device=dev, oversample=1.0)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)

    def fingerprint(c):
        nnz = c.nnz()
        p_ip, p_ix, p_dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _ffi.check(_ffi.lib.sprs_hip_csmat_device_ptrs(c._h, C.byref(p_ip), C.byref(p_ix), C.byref(p_dt)))
        blk = 1 << 27
        buf = torch.empty(blk, dtype=torch.int64, device=dev)
        w = torch.arange(1, blk + 1, dtype=torch.int64, device=dev)
        sums = []
        for base in (p_ix.value, p_dt.value):
            acc = 0
            for lo in range(0, nnz, blk):
                m = min(blk, nnz - lo)
                _ffi.check(_ffi.lib.sprs_hip_memcpy_d2d(C.c_void_p(buf.data_ptr()), C.c_void_p(base + lo * 8), m * 8, None))
                torch.cuda.synchronize()
                acc = (acc * 1000003 + int((buf[:m] * w[:m]).sum().item())) & 0xFFFFFFFFFFFFFFFF   # int64 wraps: fine for a checksum
            sums.append(acc)
        ip = torch.empty(n + 1, dtype=torch.int64, device=dev)
        _ffi.check(_ffi.lib.sprs_hip_memcpy_d2d(C.c_void_p(ip.data_ptr()), p_ip, (n + 1) * 8, None))
        torch.cuda.synchronize()
        return nnz, int((ip * torch.arange(1, n + 2, dtype=torch.int64, device=dev)).sum().item()), sums[0], sums[1]

    prints = []
    for _ in range(5):
        c = smmp.mul_csr_csr(a, a)
        prints.append(fingerprint(c))
        del c
    assert prints[0][0] > 3_000_000_000
    assert all(p == prints[0] for p in prints[1:]), prints


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_kept_plan_symbolic_numeric_split(hip, idx, ptr):
    """sprs_hip_spgemm_plan_*: the symbolic phase kept between smmp::symbolic and smmp::numeric (smmp.rs:81-131,
    151-189).  plan.product() == a * b bit for bit; plan.structure() == symbolic(a, b); plan.numeric(c) fills the values
    of a c with that structure — also after the VALUES of a changed in place — and leaves c's indices alone; a plan
    refuses other operands and a c of another structure."""
    from sprs_amd import gen, smmp
    from sprs_amd.device import DeviceCsMat
    n = 20000
    indptr, indices, data = gen.rmat_csr(n, 10, seed=21)
    ip, ix, dt = indptr.numpy().astype(ptr), indices.numpy().astype(idx), data.numpy()
    a = DeviceCsMat.from_host((n, n), ip, ix, dt)
    full = (a * a).to_host()
    plan = smmp.SpgemmPlan(a, a)
    assert plan.nnz() == full[3].size
    for x, y in zip(full[1:], plan.product().to_host()[1:]):
        assert np.array_equal(x, y)
    st = plan.structure()
    sh, sip, six, sdt = st.to_host()
    assert np.array_equal(sip, full[1]) and np.array_equal(six, full[2]) and not sdt.any()
    plan.numeric(st)                                             # value kernels only
    assert np.array_equal(st.to_host()[3], full[3]) and np.array_equal(st.to_host()[2], full[2])
    # new values, same structure: a second matrix sharing nothing but the pattern gives the reference result
    dt2 = dt * 3.0 + 1.0
    a2 = DeviceCsMat.from_host((n, n), ip, ix, dt2)
    want = (a2 * a2).to_host()
    with pytest.raises(hip.SprsHipError) as e:                   # the plan belongs to (a, a)
        plan_for_other(hip, plan, a2, st)
    assert e.value.status == hip._ffi.INVALID_ARG
    plan2 = smmp.SpgemmPlan(a2, a2)
    plan2.numeric(st)                                            # same structure, other values
    assert np.array_equal(st.to_host()[3], want[3])
    other = (a2 * DeviceCsMat.eye(n, idx, ptr))                  # a matrix with another structure
    with pytest.raises(hip.SprsHipError) as e:
        plan2.numeric(other)
    assert e.value.status == hip._ffi.BAD_STRUCTURE


def test_numeric_drops_every_cached_copy_of_c(hip):
    """sprs_hip_spgemm_numeric / _plan_numeric rewrite C's values in place.  A dense . sparse product had left
    a CSC copy of C (and a transpose view with its plans) in the handle: after the numeric call those must be rebuilt from the
    new values, for both entries, and the repeated dense . sparse product reuses ONE view."""
    from sprs_amd import gen, prod, smmp
    from sprs_amd.device import DeviceCsMat
    from sprs_amd.prod import DeviceMat
    n, k = 3000, 5
    indptr, indices, data = gen.rmat_csr(n, 6, seed=31)
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    a = DeviceCsMat.from_host((n, n), ip, ix, dt)
    a2 = DeviceCsMat.from_host((n, n), ip, ix, dt * 2.0 + 0.25)
    rng = np.random.default_rng(5)
    lhs = rng.random((k, n)) + 0.5
    dl = DeviceMat.from_host(lhs)

    def dense_of(c):
        sh, cip, cix, cdt = c.to_host()
        m = np.zeros(sh)
        for r in range(sh[0]):
            m[r, cix[cip[r]:cip[r + 1]].astype(np.int64)] = cdt[cip[r]:cip[r + 1]]
        return m

    c = a * a
    first = prod.dense_dot_csmat(dl, c).to_host()
    assert np.allclose(first, lhs @ dense_of(c), rtol=1e-10, atol=0)
    again = prod.dense_dot_csmat(dl, c).to_host()                # the kept view and its plans
    assert np.array_equal(first, again)
    smmp.numeric(a2, a2, c)                                      # same structure, new values, in place
    want = lhs @ dense_of(c)
    got = prod.dense_dot_csmat(dl, c).to_host()
    assert np.allclose(got, want, rtol=1e-10, atol=0) and not np.allclose(got, first, rtol=1e-6, atol=0)
    plan = smmp.SpgemmPlan(a, a)
    plan.numeric(c)                                              # back to the first values through the kept plan
    back = prod.dense_dot_csmat(dl, c).to_host()
    assert np.array_equal(back, first)


def plan_for_other(hip, plan, a2, c):
    import ctypes as C
    from sprs_amd._ffi import check, lib
    check(lib.sprs_hip_spgemm_plan_numeric(plan._h, a2._h, a2._h, c._h))


def test_task_order_does_not_change_the_result(hip):
    """window-major (default) and row-major order of the large-row tasks: same bits (the order only decides which
    tasks run side by side)"""
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat
    n = 30000
    indptr, indices, data = gen.rmat_csr(n, 12, seed=5)
    a = DeviceCsMat.from_host((n, n), indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy())
    r1 = (a * a).to_host()
    hip.set_option("spgemm_task_order", 2)
    try:
        r2 = (a * a).to_host()
    finally:
        hip.set_option("spgemm_task_order", 0)
    for x, y in zip(r1[1:], r2[1:]):
        assert np.array_equal(x, y)


def test_unordered_adds_option(hip):
    """option spgemm_ordered = 0 (supported opt-out of the reference's order of additions in the large-row kernel): the
    structure stays bit-identical; every value is the sum of the same products, so it agrees with the ordered result to the
    rounding of a reordered sum — |diff| <= n_products * eps * sum |products| per entry — and, on an R-MAT product of
    positive values (no cancellation), to 1e-10 relative, the north star's tolerance.  The default is restored."""
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat
    cases = [(_order_stress_cases()[i]) for i in (0, 1, 2, 3)]
    eps = np.finfo(np.float64).eps
    for a, b in cases:
        A = DeviceCsMat.from_host(*a)
        B = DeviceCsMat.from_host(*b)
        exact = (A * B).to_host()
        hip.set_option("spgemm_ordered", 0)
        try:
            relaxed = (A * B).to_host()
        finally:
            hip.set_option("spgemm_ordered", 1)
        assert np.array_equal(exact[1], relaxed[1]) and np.array_equal(exact[2], relaxed[2])
        # bound from |A| * |B| through the same kernels (ordered)
        absA = DeviceCsMat.from_host(a[0], a[1], a[2], np.abs(a[3]))
        absB = DeviceCsMat.from_host(b[0], b[1], b[2], np.abs(b[3]))
        mag = (absA * absB).to_host()[3]
        n_prod = int(np.diff(a[1].astype(np.int64)).max())
        assert np.all(np.abs(exact[3] - relaxed[3]) <= 2.0 * n_prod * eps * mag)
    n = 30000
    indptr, indices, data = gen.rmat_csr(n, 12, seed=5)
    m = DeviceCsMat.from_host((n, n), indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy())
    exact = (m * m).to_host()
    hip.set_option("spgemm_ordered", 0)
    try:
        relaxed = (m * m).to_host()
    finally:
        hip.set_option("spgemm_ordered", 1)
    assert np.array_equal(exact[1], relaxed[1]) and np.array_equal(exact[2], relaxed[2])
    assert np.max(np.abs(exact[3] - relaxed[3]) / np.maximum(np.abs(exact[3]), 1e-300)) <= 1e-10
    assert hip.get_option("spgemm_ordered") == 1
