"""GPU parity tests of the BANDED SpMV plan (sprs_amd/csrc/spmv_band.hip: hot columns served from LDS,
16-bit local column ids, compact row lists per piece) against the CPU oracle — the same bar as
test_spmv_gpu.py: <= 1e-10 relative (north star), empty rows exact.  Reference semantics:
prod::mul_acc_mat_vec_csr, sprs/src/sparse/prod.rs:103-127.

The shapes below are chosen to hit the plan's corner cases at sizes the oracle handles in seconds:
rows spanning several hot tiles, ranges and workgroups (register and head carries), tiles full of row starts,
hot slices without entries, a cold rest in several label ranges, every index-width combination."""
import numpy as np
import pytest

from conftest import IDX_COMBOS
from helpers import ragged_csr, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def hip():
    import sprs_amd
    if sprs_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (no CPU fallback exists)")
    return sprs_amd


class band_options:
    def __init__(self, hip, hot, phases, rounds=0, split=0, tile=0, cold_tiles=0, hot_run=0, overlap=0, tail=0):
        self.hip, self.vals = hip, dict(spmv_band=1, spmv_band_hot=hot, spmv_band_phases=phases, spmv_band_rounds=rounds,
                                        spmv_band_split=split, spmv_band_tile=tile, spmv_band_cold_tiles=cold_tiles,
                                        spmv_band_hot_run=hot_run, spmv_band_overlap=overlap, spmv_band_tail=tail)

    def __enter__(self):
        for k, v in self.vals.items():
            self.hip.set_option(k, v)

    def __exit__(self, *exc):
        for k in self.vals:
            self.hip.set_option(k, 0)


def oracle_spmv(shape, ip, ix, dt, x, y=None):
    from oracle import oracle
    out = np.zeros(shape[0]) if y is None else y.copy()
    oracle.mul_acc_mat_vec_csr(shape, ip.astype(np.uint64), ix.astype(np.uint64), dt, x, out)   # same arithmetic at any width
    return out


def check_band(hip, shape, ip, ix, dt, seed=0, expect_kind=3):
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    rng = np.random.default_rng(seed)
    x = rng.random(shape[1]) + 0.5
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    xv = DeviceVec.from_host(x)
    y = (a * xv).to_host()
    assert a.spmv_plan_info()[0] == expect_kind
    ref = oracle_spmv(shape, ip, ix, dt, x)
    assert rel_err(y, ref) <= TOL
    empty = np.diff(ip.astype(np.int64)) == 0
    assert np.all(y[empty] == 0.0)
    assert np.array_equal(y, (a * xv).to_host())            # cached plan + scratch: bit-identical
    y0 = rng.random(shape[0]) + 0.5
    yv = DeviceVec.from_host(y0)
    prod.mul_acc_mat_vec_csr(a, xv, yv)                     # accumulate form (prod.rs:120-126)
    y2 = yv.to_host()
    assert rel_err(y2, oracle_spmv(shape, ip, ix, dt, x, y=y0)) <= TOL
    assert np.array_equal(y2[empty], y0[empty])             # empty rows untouched, bit for bit
    return y


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS + [(np.uint64, np.uint32)])
@pytest.mark.parametrize("hot,phases,tile", [(2, 1, 16384), (5, 3, 8192)])
def test_rmat_vs_oracle(hip, idx, ptr, hot, phases, tile):
    from sprs_amd import gen
    n = 50000
    indptr, indices, data = gen.rmat_csr(n, 16)
    ip, ix, dt = indptr.numpy().astype(ptr), indices.numpy().astype(idx), data.numpy()
    with band_options(hip, hot, phases, tile=tile):
        check_band(hip, (n, n), ip, ix, dt)


def test_hub_rows_and_many_segments(hip):
    """3 dense rows (20 000 entries each: every hot slice sees rows that span several tiles, ranges and workgroups ->
    register carries inside a range, head carries between ranges, runs of ranges without a row start) and 7 000 rows of
    33 entries (tiles in which every lane holds several row starts), short and empty rows in between; one to forty
    hot workgroups per CU, ranges of one tile up to a whole segment, one to seven tiles per cold range."""
    lens = [0, 20000, 3, 0] + [33] * 3500 + [20000] + [1, 0, 31, 32] * 50 + [33] * 3500 + [20000, 0, 0, 7]
    shape, ip, ix, dt = ragged_csr(lens, 20000, seed=3)
    for rounds, tile, ct, run in ((1, 8192, 4, 0), (3, 16384, 1, 1), (40, 8192, 7, 3), (2, 16384, 2, 1000)):
        with band_options(hip, 2, 2, rounds=rounds, tile=tile, cold_tiles=ct, hot_run=run):
            check_band(hip, shape, ip, ix, dt, seed=rounds)


def test_reduction_beside_the_short_rows(hip):
    """the two streams forced on a small matrix: the reduction of the long rows starts when the hot slices and the cold
    pieces are done and runs beside the short rows (whose carries follow them on the second stream) — the same additions in
    the same order as with the whole second stream joined first (spmv_band_tail = 2) and as on one stream: bit-identical"""
    from sprs_amd.device import DeviceCsMat, DeviceVec
    rng = np.random.default_rng(21)
    rows, cols = 6000, 300000
    lens = rng.integers(1, 60, size=rows)
    lens[::97] = 5000                                          # hub rows: runs that span ranges in the early slices
    lens[5::7] = 2500                                          # short rows (below the split) that span several cold tiles: carries into y
    lens[3::50] = 0
    shape, ip, ix, dt = ragged_csr(list(lens), cols, seed=22)
    x = rng.random(cols) + 0.5
    out = {}
    for overlap, tail in ((2, 0), (1, 0), (1, 2)):
        with band_options(hip, 12, 1, tile=8192, rounds=3, hot_run=2, split=3000, cold_tiles=1, overlap=overlap, tail=tail):
            check_band(hip, shape, ip, ix, dt, seed=4)
            a = DeviceCsMat.from_host(shape, ip, ix, dt)
            xv = DeviceVec.from_host(x)
            out[(overlap, tail)] = [(a * xv).to_host() for _ in range(3)]
    first = out[(2, 0)][0]
    for ys in out.values():
        for y in ys:
            assert np.array_equal(y, first)


def test_split_and_empty_pieces(hip):
    """all long rows live in a narrow column range: most hot slices and hash pieces have no entries; a second
    matrix has no short rows at all, a third no long rows (the banded plan does not apply: plain tiles)"""
    rng = np.random.default_rng(11)
    n, cols = 3000, 60000
    lens = rng.integers(40, 90, size=n)
    ip = np.zeros(n + 1, dtype=np.uint64)
    ip[1:] = np.cumsum(lens)
    ix = np.concatenate([np.sort(rng.choice(700, size=l, replace=False)) + 17000 for l in lens]).astype(np.uint64)
    dt = rng.random(ix.size) + 0.5
    with band_options(hip, 6, 2):
        check_band(hip, (n, cols), ip, ix, dt)
    with band_options(hip, 3, 1, split=2):
        shape, ip2, ix2, dt2 = ragged_csr([5, 9, 2, 64, 300] * 400, 30000, seed=5)
        check_band(hip, shape, ip2, ix2, dt2)
    with band_options(hip, 3, 1, split=1000):
        shape, ip3, ix3, dt3 = ragged_csr([5, 9, 0, 64, 300] * 100, 9000, seed=6)
        check_band(hip, shape, ip3, ix3, dt3, expect_kind=1)


def test_band_equals_sliced_and_plain_within_rounding(hip):
    """the three plans group the products of a row differently: equal to the oracle within tolerance, to each
    other within rounding"""
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat, DeviceVec
    n = 40000
    indptr, indices, data = gen.rmat_csr(n, 24, seed=9)
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    x = gen.dense_vector(n, seed=2).numpy()
    out = {}
    try:
        for name, opts in (("band", dict(spmv_band=1, spmv_band_hot=3)), ("sliced", dict(spmv_band=2, spmv_xcs=1)),
                           ("plain", dict(spmv_band=2, spmv_xcs=2))):
            for k, v in opts.items():
                hip.set_option(k, v)
            a = DeviceCsMat.from_host((n, n), ip, ix, dt)
            out[name] = (a * DeviceVec.from_host(x)).to_host()
            assert a.spmv_plan_info()[0] == {"band": 3, "sliced": 2, "plain": 1}[name]
    finally:
        for k in ("spmv_band", "spmv_band_hot", "spmv_xcs"):
            hip.set_option(k, 0)
    ref = oracle_spmv((n, n), ip, ix, dt, x)
    for name, y in out.items():
        assert rel_err(y, ref) <= TOL, name
    assert rel_err(out["band"], out["plain"]) <= 1e-13 and rel_err(out["band"], out["sliced"]) <= 1e-13


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_shapes_and_plan_options(hip, seed):
    """seeded random ragged matrices (empty rows, rows of one entry, rows around the split, hub rows, a column range nobody
    references) under random plan geometries — tile size, slices, phases, rounds, range lengths, cold range lengths, the
    two-part reduction — all against the oracle, the accumulate form and the bit-identical repeat included (check_band)"""
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.integers(200, 1500))
    cols = int(rng.integers(9000, 70000))
    kinds = rng.integers(0, 6, size=rows)
    lens = np.where(kinds == 0, 0, np.where(kinds == 1, 1, np.where(kinds == 2, rng.integers(20, 30, size=rows),
                    np.where(kinds == 3, rng.integers(2, 24, size=rows), rng.integers(24, 200, size=rows)))))
    hubs = rng.choice(rows, size=3, replace=False)
    lens[hubs] = rng.integers(2000, min(cols, 9000), size=3)
    shape, ip, ix, dt = ragged_csr([int(v) for v in lens], cols, seed=seed)
    ix = (ix.astype(np.int64) * 7 // 8).astype(ix.dtype)                 # the top eighth of the columns is never referenced ...
    for r in range(rows):                                                # ... (keep the rows strictly increasing)
        seg = ix[int(ip[r]):int(ip[r + 1])]
        if seg.size and np.any(np.diff(seg.astype(np.int64)) <= 0):
            ix[int(ip[r]):int(ip[r + 1])] = np.unique(seg)[:1].repeat(seg.size) + np.arange(seg.size, dtype=ix.dtype)
    ix = np.minimum(ix, cols - 1)
    ok = all(np.all(np.diff(ix[int(ip[r]):int(ip[r + 1])].astype(np.int64)) > 0) for r in range(rows))
    if not ok:
        pytest.skip("degenerate draw")
    tile = int(rng.choice([8192, 16384]))
    hot = int(rng.integers(1, max(2, cols // tile + 1)))
    opts = dict(rounds=int(rng.integers(1, 6)), tile=tile, cold_tiles=int(rng.integers(1, 6)), hot_run=int(rng.integers(1, 6)),
                split=int(rng.choice([2, 8, 24, 40])))
    with band_options(hip, hot, int(rng.integers(1, 4)), **opts):
        check_band(hip, shape, ip, ix, dt, seed=seed)
