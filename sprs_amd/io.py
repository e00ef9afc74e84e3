"""Matrix Market coordinate files <-> triplets: host-side twin of sprs::io
(sprs/src/io.rs:84-111 parse_header, :139-278 read_matrix_market_from_bufread, :309-360
write_matrix_market_to_bufwrite).  The text stays on the host, as in the reference; assembling the
triplets into CSR / CSC happens on the device (`TriMat.to_csr` / `.to_csc`, triplet.py).
Scalars here are f64 ("real" / "integer" files; "pattern" files when asked for, with values 1.0);
complex files raise the reference's mismatch error."""
import io as _io

import numpy as np

from .triplet import TriMat


class IoError(Exception):
    """sprs::io::IoError (io.rs:12-23); `kind` is the variant name."""

    def __init__(self, kind, message):
        self.kind = kind
        super().__init__(message)


def _bad():
    return IoError("BadMatrixMarketFile", "Bad matrix market file.")


def parse_header(header):
    """io.rs:84-111 on the lower-cased first line -> (symmetry, data type)."""
    if not header.startswith("%%matrixmarket matrix coordinate"):
        raise _bad()
    for key in ("real", "integer", "complex", "pattern"):
        if key in header:
            data_type = key
            break
    else:
        raise _bad()
    if "general" in header:
        sym = "general"
    elif "skew-symmetric" in header:
        sym = "skew-symmetric"
    elif "symmetric" in header:
        sym = "symmetric"
    elif "hermitian" in header:
        sym = "hermitian"
    else:
        raise _bad()
    return sym, data_type


def read_matrix_market(source, kind="real"):
    """read_matrix_market / read_matrix_market_from_bufread (io.rs:118-278) -> TriMat.

    source: a path, or any iterable of lines / text stream.  kind: the scalar the caller wants —
    "real" (f64; the file must be real), "integer" (the file must be integer; values kept as f64) or
    "pattern" (any file; values dropped, 1.0 stored).  Symmetric / skew-symmetric / hermitian files are
    expanded as the reference does (the mirrored entry follows its original; a hermitian real entry
    mirrors unchanged)."""
    if isinstance(source, (str, bytes)) and "\n" not in str(source):
        with open(source, "r") as f:
            return read_matrix_market(f, kind)
    if isinstance(source, str):
        source = _io.StringIO(source)
    lines = iter(source)
    try:
        header = next(lines).lower()
    except StopIteration:
        raise _bad()
    sym, data_type = parse_header(header)
    file_kind = {"integer": "integer", "real": "real", "complex": "complex", "pattern": "pattern"}[data_type]
    if kind != "pattern" and kind != file_kind:
        raise IoError("MismatchedMatrixMarketRead",
                      "Tried to load %s file into %s matrix." % (file_kind, kind))          # io.rs:33-39
    dropping = kind == "pattern" and file_kind != "pattern"
    # comment lines after the header (io.rs:176-184: only lines starting with '%' are skipped — a blank line is
    # taken as the size line and fails there; at end of file the reference would spin, here it is a bad file)
    for line in lines:
        if not line.startswith("%"):
            break
    else:
        raise _bad()
    infos = []
    for tok in line.split():
        try:
            infos.append(int(tok))
        except ValueError:
            pass                                          # filter_map(parse().ok()), io.rs:190-192
    if len(infos) != 3 or min(infos) < 0:
        raise _bad()
    rows, cols, entries = infos
    r_out, c_out, v_out = [], [], []
    for _ in range(entries):
        while True:                                       # skip all-whitespace lines (io.rs:213-222)
            try:
                line = next(lines)
            except StopIteration:
                line = ""
            if line != "" and line.strip() == "":
                continue
            break
        toks = line.split()
        try:
            row, col = int(toks[0]), int(toks[1])
            if row < 0 or col < 0:
                raise ValueError
        except (IndexError, ValueError):
            raise _bad()
        if row == 0 or col == 0:                          # 1-based (checked_sub, io.rs:240-241)
            raise _bad()
        row -= 1
        col -= 1
        rest = toks[2:]
        if file_kind == "pattern" or kind == "pattern":
            val = 1.0
            used = 0
        else:
            try:
                val = float(int(rest[0])) if file_kind == "integer" else float(rest[0])
            except (IndexError, ValueError):
                raise _bad()
            used = 1
        r_out.append(row)
        c_out.append(col)
        v_out.append(val)
        if sym != "general" and row != col:
            r_out.append(col)
            c_out.append(row)
            v_out.append(-val if sym == "skew-symmetric" else val)
        if sym == "skew-symmetric" and row == col:
            raise _bad()
        if dropping:
            if len(rest) == 0:                            # the file has data, it must be there (io.rs:266-270)
                raise _bad()
        elif len(rest) != used:                           # all data must be consumed (io.rs:271-275)
            raise _bad()
    if r_out and (max(r_out) >= rows or max(c_out) >= cols):
        raise _bad()
    return TriMat((rows, cols), r_out, c_out, v_out)


def write_matrix_market(dest, mat, kind="real"):
    """write_matrix_market (io.rs:294-360): header, a comment line, `rows cols nnz`, then one
    `row col value` line per stored entry, 1-based.  mat: TriMat, or a DeviceCsMat (downloaded; entries in
    storage order)."""
    if isinstance(mat, TriMat):
        rows, cols = mat.shape()
        trip = zip(mat.row_inds.tolist(), mat.col_inds.tolist(), mat.data.tolist())
        nnz = mat.nnz()
    else:
        (rows, cols), indptr, indices, data = mat.to_host()
        outer = np.repeat(np.arange(indptr.size - 1), np.diff(indptr.astype(np.int64)))
        r, c = (outer, indices) if mat.is_csr() else (indices, outer)
        trip = zip(r.tolist(), c.tolist(), data.tolist())
        nnz = int(data.size)
    own = isinstance(dest, (str, bytes))
    f = open(dest, "w") if own else dest
    try:
        f.write("%%%%MatrixMarket matrix coordinate %s general\n" % kind)
        f.write("% written by sprs_amd\n")
        f.write("%d %d %d\n" % (rows, cols, nnz))
        for r, c, v in trip:
            f.write("%d %d %s\n" % (r + 1, c + 1, repr(int(v)) if kind == "integer" else repr(float(v))))
    finally:
        if own:
            f.close()
