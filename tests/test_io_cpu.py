"""Host side of the Matrix Market path (sprs_amd/io.py, twin of sprs/src/io.rs) and the oracle's
restatement of TriMatIter::into_cs — no GPU needed: the parser produces host triplets."""
import io

import numpy as np
import pytest

SIMPLE = """%%MatrixMarket matrix coordinate real general
%=================================================================================
% comment lines, then `rows cols entries`, then one entry per non-empty line (1-based)
%=================================================================================
  5  5  8
    1     1   1.000e+00
    2     2   1.050e+01

    3     3   1.500e-02
    \t
    1     4   6.000e+00
    4     2   2.505e+02
    4     4  -2.800e+02
    4     5   3.332e+01
    5     5   1.200e+01
"""


def test_simple_matrix_market_read():
    # io.rs:477-491 simple_matrix_market_read: the asserted triplets of data/matrix_market/simple.mm
    from sprs_amd.io import read_matrix_market
    m = read_matrix_market(SIMPLE)
    assert m.shape() == (5, 5) and m.nnz() == 8
    assert m.row_inds.tolist() == [0, 1, 2, 0, 3, 3, 3, 4]
    assert m.col_inds.tolist() == [0, 1, 2, 3, 1, 3, 4, 4]
    assert m.data.tolist() == [1., 10.5, 1.5e-02, 6., 2.505e2, -2.8e2, 3.332e1, 1.2e+1]
    again = read_matrix_market(io.StringIO(SIMPLE))                   # from a stream (io.rs:607-624)
    assert again.data.tolist() == m.data.tolist()


def test_kinds_and_mismatch_errors():
    # io.rs:493-533 failing_matrix_market_reads + :627-637 int_matrix_market_read
    from sprs_amd.io import IoError, read_matrix_market
    int_file = SIMPLE.replace("real", "integer").replace("1.000e+00", "1").replace("1.050e+01", "1") \
        .replace("1.500e-02", "1").replace("6.000e+00", "6").replace("2.505e+02", "2") \
        .replace("-2.800e+02", "-2").replace("3.332e+01", "3").replace("1.200e+01", "1")
    assert read_matrix_market(int_file, kind="integer").data.tolist() == [1, 1, 1, 6, 2, -2, 3, 1]
    with pytest.raises(IoError) as e:
        read_matrix_market(int_file, kind="real")
    assert e.value.kind == "MismatchedMatrixMarketRead" and str(e.value) == "Tried to load integer file into real matrix."
    with pytest.raises(IoError) as e:
        read_matrix_market(SIMPLE.replace("real", "complex"), kind="real")
    assert str(e.value) == "Tried to load complex file into real matrix."
    with pytest.raises(IoError):
        read_matrix_market(SIMPLE, kind="integer")
    # any file can be read as a pattern (io.rs:162-169, 852-...): values dropped
    pat = read_matrix_market(SIMPLE, kind="pattern")
    assert pat.nnz() == 8 and pat.data.tolist() == [1.0] * 8
    real_pattern_file = "%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 2\n3 1\n"
    p2 = read_matrix_market(real_pattern_file, kind="pattern")
    assert (p2.row_inds.tolist(), p2.col_inds.tolist()) == ([0, 2], [1, 0])


@pytest.mark.parametrize("body", [
    "2 2 1\n1 2 3.5 7.0\n",              # too many elements in an entry (io.rs:640-645)
    "2 2 3\n1 2 3.5\n",                   # not enough entries (io.rs:648-653)
    "2 2 1\n0 1 3.5\n",                   # indices are 1-based
    "2 2\n1 1 1.0\n",                     # size line too short
    "2 2 1 9\n1 1 1.0\n",                 # size line too long
    "2 2 1\n1 1\n",                       # missing value
    "\n2 2 1\n1 1 1.0\n",                # a blank line where the size line is expected
])
def test_bad_files(body):
    from sprs_amd.io import IoError, read_matrix_market
    with pytest.raises(IoError) as e:
        read_matrix_market("%%MatrixMarket matrix coordinate real general\n% c\n" + body)
    assert e.value.kind == "BadMatrixMarketFile" and str(e.value) == "Bad matrix market file."
    with pytest.raises(IoError):
        read_matrix_market("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")      # not coordinate
    with pytest.raises(IoError):
        read_matrix_market("%%MatrixMarket matrix coordinate real\n2 2 0\n")                        # no symmetry


def test_symmetry_modes_expand_like_the_reference():
    # io.rs:243-264: the mirrored entry follows its original; diagonal entries are not mirrored;
    # a skew-symmetric file may not hold a diagonal entry
    from sprs_amd.io import IoError, read_matrix_market
    body = "3 3 3\n1 1 2.0\n2 1 -1.5\n3 2 4.0\n"
    s = read_matrix_market("%%MatrixMarket matrix coordinate real symmetric\n" + body)
    assert s.row_inds.tolist() == [0, 1, 0, 2, 1] and s.col_inds.tolist() == [0, 0, 1, 1, 2]
    assert s.data.tolist() == [2.0, -1.5, -1.5, 4.0, 4.0]
    h = read_matrix_market("%%MatrixMarket matrix coordinate real hermitian\n" + body)
    assert h.data.tolist() == s.data.tolist()
    k = read_matrix_market("%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 -1.5\n3 2 4.0\n")
    assert k.data.tolist() == [-1.5, 1.5, 4.0, -4.0] and k.row_inds.tolist() == [1, 0, 2, 1]
    with pytest.raises(IoError):
        read_matrix_market("%%MatrixMarket matrix coordinate real skew-symmetric\n" + body)


def test_against_scipy_mmread(tmp_path):
    """independent cross-check: scipy.io.mmread of the same text (dense comparison)"""
    import scipy.io
    from oracle import oracle
    from sprs_amd.io import read_matrix_market, write_matrix_market
    from sprs_amd.triplet import TriMat
    rng = np.random.default_rng(0)
    n, m, k = 40, 30, 300
    t = TriMat((n, m), rng.integers(0, n, k), rng.integers(0, m, k), rng.standard_normal(k))   # with duplicates
    path = tmp_path / "rand.mtx"
    write_matrix_market(str(path), t)
    back = read_matrix_market(str(path))
    assert back.row_inds.tolist() == t.row_inds.tolist() and back.data.tolist() == t.data.tolist()   # repr round trip
    ref = scipy.io.mmread(str(path)).toarray()
    ip, ix, dt = oracle.triplets_to_cs((n, m), back.row_inds, back.col_inds, back.data)
    dense = np.zeros((n, m))
    for i in range(n):
        dense[i, ix[int(ip[i]):int(ip[i + 1])].astype(int)] = dt[int(ip[i]):int(ip[i + 1])]
    assert np.allclose(dense, ref, rtol=1e-13, atol=1e-13)


def test_oracle_into_cs_semantics():
    """triplet_iter.rs:127-224: sorted rows, duplicates folded left to right, zeros kept, empty outers,
    CSC twin; triplet.rs tests `triplet_unordered` / `triplet_additions` style cases"""
    from oracle import oracle
    r, c = [2, 0, 2, 0, 2, 1], [1, 3, 1, 0, 1, 2]
    v = [1e16, 5.0, 1.0, 0.0, -1e16, 7.0]
    ip, ix, dt = oracle.triplets_to_cs((4, 4), r, c, v)
    assert ip.tolist() == [0, 2, 3, 4, 4] and ix.tolist() == [0, 3, 2, 1]
    assert dt.tolist() == [0.0, 5.0, 7.0, (1e16 + 1.0) + -1e16]         # folded in triplet order, explicit zero kept
    ipc, ixc, dtc = oracle.triplets_to_cs((4, 4), r, c, v, storage="CSC")
    assert ipc.tolist() == [0, 1, 2, 3, 4] and ixc.tolist() == [0, 2, 1, 0]
    e = oracle.triplets_to_cs((3, 2), [], [], [])
    assert e[0].tolist() == [0, 0, 0, 0] and e[1].size == 0
