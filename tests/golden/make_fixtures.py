#!/usr/bin/env python3
"""Extract the reference's golden vectors for the SpMV / SpGEMM path into
tests/golden/sprs_fixtures.json.

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):   python tests/golden/make_fixtures.py

Sources (numbers only — no reference code is copied):
  sprs/src/test_data.rs:6-124      mat1..mat5, expected products, dense mats
  sprs/src/sparse/prod.rs:325-423  mul_csc_vec / mul_csr_vec known answers
  sprs/src/sparse/prod.rs:502-542  mul_csr_dense_rowmaj expected (mat5 x mat_dense2)
  sprs/src/sparse/smmp.rs:475-489  mul_zero_rows
  sprs/src/sparse/smmp.rs:515-555  mul_complex (structure part)
  sprs/tests/block_matrix.rs:71-108 block product structure
  sprs/src/sparse/vec.rs:1648-1673 dot_product known answer
  sprs/src/sparse/linalg/bicgstab.rs:336-369  test_bicgstab_f64 system (CSC 4x4, tol, max_iter)
  sprs/src/sparse/triplet.rs:343-646  triplet_incremental / _unordered / _additions / _from_vecs / _mutate_entry /
                                      _to_csr / _complex / _empty_lines: triplets in insertion order + expected CSC
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sprs_fixtures.json")


def nums(s):
    return [float(t) if re.search(r"[.eE]", t) else int(t)
            for t in re.findall(r"-?\d+\.?\d*(?:[eE]-?\d+)?", s)]


def fn_body(src, name):
    m = re.search(r"pub fn %s\(\)[^{]*\{" % re.escape(name), src)
    assert m, name
    i = m.end()
    depth = 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.end():i - 1]


def csmat(src, name):
    b = fn_body(src, name)
    arrs = {k: nums(re.search(r"let %s\s*=\s*vec!\[(.*?)\];" % k, b, re.S).group(1))
            for k in ("indptr", "indices", "data")}
    ctor = re.search(r"CsMat::(new_csc|new)\(\((\d+),\s*(\d+)\)", b)
    return dict(storage="CSC" if ctor.group(1) == "new_csc" else "CSR",
                shape=[int(ctor.group(2)), int(ctor.group(3))],
                indptr=arrs["indptr"], indices=arrs["indices"],
                data=[float(v) for v in arrs["data"]])


def dense(src, name):
    b = fn_body(src, name)
    rows = re.findall(r"\[([^\[\]]+)\]", re.search(r"arr2\(&\[(.*)\]\)", b, re.S).group(1))
    return [[float(v) for v in nums(r)] for r in rows]


def triplet_cases():
    """Every matrix built in the tests of triplet.rs:343-646: the triplets in INSERTION order (duplicates are summed by
    the conversion) and the CSC the reference expects (its CSR expectation is `expected.to_csr()`)."""
    src = open(os.path.join(REF, "sprs/src/sparse/triplet.rs")).read()
    tests = src[src.index("mod test {"):]
    cases = []
    for m in re.finditer(r"fn (triplet_\w+)\(\)\s*\{", tests):
        name = m.group(1)
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(tests[i], 0)
            i += 1
        body = tests[m.end():i - 1]
        # a test may build several matrices: cut at every constructor
        ctor = r"(?:with_capacity\(\s*\((\d+),\s*(\d+)\)|TriMatI?::new\(\((\d+),\s*(\d+)\)\)|from_triplets\(\((\d+),\s*(\d+)\))"
        starts = list(re.finditer(ctor, body))
        for si, sm in enumerate(starts):
            seg = body[sm.start():starts[si + 1].start() if si + 1 < len(starts) else len(body)]
            shape = [int(v) for v in sm.groups() if v is not None]
            if "from_triplets" in sm.group(0):
                pre = body[:sm.start()]
                rows = nums(re.search(r"let row_inds = vec!\[(.*?)\];", pre, re.S).group(1))
                cols = nums(re.search(r"let col_inds = vec!\[(.*?)\];", pre, re.S).group(1))
                vals = nums(re.search(r"let data = vec!\[(.*?)\];", pre, re.S).group(1))
            else:
                trip = re.findall(r"add_triplet\((\d+),\s*(\d+),\s*(-?[\d.]+)\)", seg)
                rows, cols, vals = [int(t[0]) for t in trip], [int(t[1]) for t in trip], [float(t[2]) for t in trip]
                sm2 = re.search(r"set_triplet\(locations\[0\],\s*(\d+),\s*(\d+),\s*(-?[\d.]+)\)", seg)
                if sm2:      # triplet_mutate_entry: the single entry at that position gets a new value
                    r_, c_, v_ = int(sm2.group(1)), int(sm2.group(2)), float(sm2.group(3))
                    hits = [q for q in range(len(rows)) if rows[q] == r_ and cols[q] == c_]
                    assert len(hits) == 1
                    vals[hits[0]] = v_
            em = re.search(r"new_csc\(\s*\((\d+),\s*(\d+)\),\s*vec!\[(.*?)\],\s*vec!\[(.*?)\],\s*vec!\[(.*?)\],?\s*\)", seg, re.S)
            if em:
                assert [int(em.group(1)), int(em.group(2))] == shape
                exp = dict(indptr=nums(em.group(3)), indices=nums(em.group(4)), data=[float(v) for v in nums(em.group(5))])
            else:            # expectations given as slices (triplet_empty_lines): the CSC ones come last in the segment
                ip = re.findall(r"assert_eq!\(m\.indptr\(\), &\[(.*?)\]\[\.\.\]\)", seg)
                ix = re.findall(r"assert_eq!\(m\.indices\(\), &\[(.*?)\]\)", seg)
                dt = re.findall(r"assert_eq!\(m\.data\(\), &\[(.*?)\]\)", seg)
                exp = dict(indptr=nums(ip[-1]), indices=nums(ix[-1]), data=[float(v) for v in nums(dt[-1])])
            assert len(exp["indptr"]) == shape[1] + 1 and exp["indptr"][-1] == len(exp["indices"]) == len(exp["data"])
            cases.append(dict(name="%s#%d" % (name, si), shape=shape, rows=rows, cols=cols, data=[float(v) for v in vals], csc=exp))
    assert len(cases) >= 10, len(cases)
    return cases


def main():
    td = open(os.path.join(REF, "sprs/src/test_data.rs")).read()
    prod = open(os.path.join(REF, "sprs/src/sparse/prod.rs")).read()
    fx = {"_provenance": "generated by tests/golden/make_fixtures.py from sparsemat/sprs 0.11.5"}
    for name in ("mat1", "mat1_csc", "mat2", "mat3", "mat4", "mat5", "mat1_times_2",
                 "mat1_self_matprod", "mat1_matprod_mat2", "mat1_csc_matprod_mat4"):
        fx[name] = csmat(td, name)
    fx["mat_dense1"] = dense(td, "mat_dense1")
    fx["mat_dense2"] = dense(td, "mat_dense2")

    def test_body(name):
        m = re.search(r"fn %s\(\)\s*\{" % name, prod)
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(prod[i], 0)
            i += 1
        return prod[m.end():i - 1]

    def spmv_case(name):
        b = test_body(name)
        g = lambda pat: nums(re.search(pat, b, re.S).group(1))
        return dict(shape=[5, 5],
                    indptr=g(r"let indptr: &\[usize\] = &\[(.*?)\];"),
                    indices=g(r"let indices: &\[usize\] = &\[(.*?)\];"),
                    data=g(r"let data: &\[f64\] = &\[(.*?)\];"),
                    x=[0.1, 0.2, -0.1, 0.3, 0.9],
                    expected=g(r"let expected_output =\s*(?:vec!)?\[(.*?)\];"),
                    epsilon=1e-7)

    fx["mul_csr_vec"] = spmv_case("mul_csr_vec")          # prod.rs:375-398
    fx["mul_csc_vec"] = spmv_case("mul_csc_vec")          # prod.rs:325-348
    assert "0.1, 0.2, -0.1, 0.3, 0.9" in test_body("mul_csr_vec")

    b = test_body("mul_csr_dense_rowmaj")                 # prod.rs:502-542
    exps = re.findall(r"let expected_output = arr2\(&\[(.*?)\]\);", b, re.S)
    assert len(exps) == 2
    rows_of = lambda blk: [[float(v) for v in nums(r)] for r in re.findall(r"\[([^\[\]]+)\]", blk)]
    fx["mat1_times_mat_dense1"] = dict(rows=rows_of(exps[0]), epsilon=0.0)
    fx["mat5_times_mat_dense2"] = dict(rows=rows_of(exps[1]), epsilon=1e-8)

    # smmp.rs:475-489 (issue 239), smmp.rs:515-555, block_matrix.rs:71-108, vec.rs:1648-1673
    fx["mul_zero_rows"] = dict(a_shape=[0, 11], a_indptr=[0], b_shape=[11, 11],
                               b_indptr=[0] * 12, c_shape=[0, 11], c_nnz=0)
    fx["mul_complex_structure"] = dict(shape=[4, 4], indptr=[0, 1, 1, 3, 4], indices=[1, 0, 3, 2],
                                       c_indptr=[0, 0, 0, 2, 4], c_indices=[1, 2, 0, 3])
    fx["block_matrix_structure"] = dict(a_shape=[2, 2], a_indptr=[0, 1, 3], a_indices=[1, 0, 1],
                                        b_shape=[2, 2], b_indptr=[0, 2, 2], b_indices=[0, 1],
                                        c_indptr=[0, 0, 2], c_indices=[0, 1])
    vec = open(os.path.join(REF, "sprs/src/sparse/vec.rs")).read()
    db = re.search(r"fn dot_product\(\)\s*\{(.*?)\n    \}", vec, re.S).group(1)
    assert "CsVec::new(8, vec![0, 2, 4, 6], vec![1.; 4])" in db and "assert_eq!(16., vec1.dot(&dense_vec))" in db
    fx["dot_product"] = dict(dim=8, indices=[0, 2, 4, 6], data=[1.0] * 4,
                             dense=nums(re.search(r"let dense_vec = vec!\[(.*?)\];", db, re.S).group(1)),
                             expected=16.0)
    # bicgstab.rs:336-369 test_bicgstab_f64 (the doc example :32-66 is the same system)
    bi = open(os.path.join(REF, "sprs/src/sparse/linalg/bicgstab.rs")).read()
    tb = re.search(r"fn test_bicgstab_f64\(\)\s*\{(.*?)\n    \}", bi, re.S).group(1)
    ctor = re.search(r"CsMatI::new_csc\(\s*\((\d+),\s*(\d+)\),\s*vec!\[(.*?)\],\s*vec!\[(.*?)\],\s*vec!\[(.*?)\],", tb, re.S)
    assert "vec![1.0; 4]" in tb and "vec![1.0, 1.0, 1.0, 1.0]" in tb          # b and x0 are ones
    fx["bicgstab_example"] = dict(storage="CSC", shape=[int(ctor.group(1)), int(ctor.group(2))],
                                  indptr=nums(ctor.group(3)), indices=nums(ctor.group(4)),
                                  data=[float(v) for v in nums(ctor.group(5))],
                                  tol=float(re.search(r"let tol = ([0-9.e-]+);", tb).group(1)),
                                  max_iter=int(re.search(r"let max_iter = (\d+);", tb).group(1)))
    fx["triplet_cases"] = triplet_cases()
    with open(OUT, "w") as f:
        json.dump(fx, f, indent=1, sort_keys=True)
    print("wrote", OUT, "with", len(fx) - 1, "fixtures")


if __name__ == "__main__":
    main()
