"""GPU parity tests of the HIP SpMV path (through the C ABI) against the CPU
oracle and the reference's golden vectors.  Bar: |y_gpu - y_ref| <= 1e-10 rel
(north star); exact zeros / untouched entries for empty rows."""
import ctypes as C

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


def gpu_spmv(hip, shape, ip, ix, dt, x, y=None, validate=True):
    from sprs_amd import prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    a = DeviceCsMat.from_host(shape, ip, ix, dt, validate=validate)
    xv = DeviceVec.from_host(x)
    if y is None:
        return (a * xv).to_host()
    yv = DeviceVec.from_host(y)
    prod.mul_acc_mat_vec_csr(a, xv, yv)
    return yv.to_host()


def oracle_spmv(shape, ip, ix, dt, x, y=None):
    from oracle import oracle
    out = np.zeros(shape[0]) if y is None else y.copy()
    oracle.mul_acc_mat_vec_csr(shape, ip, ix, dt, x, out)
    return out


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS + [(np.uint64, np.uint32)])
def test_golden_mul_csr_vec(hip, golden, idx, ptr):
    # sprs/src/sparse/prod.rs:375-423
    fx = golden["mul_csr_vec"]
    shape, ip, ix, dt = as_csr(fx, idx, ptr)
    x = np.array(fx["x"])
    y = gpu_spmv(hip, shape, ip, ix, dt, x)
    assert np.all(np.abs(y - np.array(fx["expected"])) < fx["epsilon"])
    assert y[1] == 0.0                                  # empty row, operator form
    y0 = np.array([1.0, -2.5, 3.0, 4.0, 5.0])
    y2 = gpu_spmv(hip, shape, ip, ix, dt, x, y=y0)      # accumulate form keeps y (prod.rs:121)
    assert np.all(np.abs(y2 - (y0 + np.array(fx["expected"]))) < fx["epsilon"])
    assert y2[1] == -2.5


def test_config1_eye_1000(hip):
    # BASELINE config 1: CsMat::eye(1000) * dense vec == vec bit for bit (csmat.rs:406-426)
    from sprs_amd.device import DeviceCsMat, DeviceVec
    x = 0.5 + np.arange(1000) / 1000.0
    y = (DeviceCsMat.eye(1000) * DeviceVec.from_host(x)).to_host()
    assert np.array_equal(x, y)
    y = (DeviceCsMat.eye(1000, np.uint32, np.uint32) * DeviceVec.from_host(x)).to_host()
    assert np.array_equal(x, y)


@pytest.mark.parametrize("xcs,idx32,srt", [(1, 1, 1), (1, 0, 1), (1, 1, 0), (2, 1, 1)],
                         ids=["xcd-sliced-idx32-sorted", "xcd-sliced-sorted", "xcd-sliced-idx32", "plain"])
@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_rmat_vs_oracle(hip, idx, ptr, xcs, idx32, srt):
    import torch
    from sprs_amd import gen
    hip.set_option("spmv_xcs", xcs)       # 1: force the XCD-sliced plan, 2: plain tile kernel only
    hip.set_option("spmv_xcs_idx32", idx32)
    hip.set_option("spmv_sort_tiles", srt)
    try:
        n = 60000
        indptr, indices, data = gen.rmat_csr(n, 16)
        ip, ix, dt = indptr.numpy().astype(ptr), indices.numpy().astype(idx), data.numpy()
        x = gen.dense_vector(n).numpy()
        y = gpu_spmv(hip, (n, n), ip, ix, dt, x)
        ref = oracle_spmv((n, n), ip, ix, dt, x)
        assert rel_err(y, ref) <= TOL
        empty = np.diff(ip.astype(np.int64)) == 0
        assert empty.any() and np.all(y[empty] == 0.0)
        rng = np.random.default_rng(5)
        y0 = rng.random(n) + 0.5
        y2 = gpu_spmv(hip, (n, n), ip, ix, dt, x, y=y0)
        ref2 = oracle_spmv((n, n), ip, ix, dt, x, y=y0)
        assert rel_err(y2, ref2) <= TOL
        assert np.array_equal(y2[empty], y0[empty])     # empty rows untouched, bit for bit
    finally:
        hip.set_option("spmv_xcs", 0)
        hip.set_option("spmv_xcs_idx32", 1)
        hip.set_option("spmv_sort_tiles", 0)


@pytest.mark.parametrize("idx,ptr", IDX_COMBOS)
def test_column_relabelling(hip, idx, ptr):
    """The sliced plan renumbers the columns by popularity class and permutes x per SpMV (spmv.hip rl_*).
    Which x-slice an entry falls into follows its label, so the 8 partial sums of a row group the same
    products differently: equal to the oracle within tolerance either way, to each other within rounding,
    and bit-identical from one SpMV to the next on the same handle (cached plan + scratch); accumulate
    form and columns nobody references included."""
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    n = 70000
    indptr, indices, data = gen.rmat_csr(n, 24, seed=17)
    a_host = ((n, n), indptr.numpy().astype(ptr), indices.numpy().astype(idx), data.numpy())
    x = gen.dense_vector(n, seed=5).numpy()
    ref = oracle_spmv(*a_host, x)
    out = {}
    try:
        hip.set_option("spmv_xcs", 1)
        for relabel in (1, 2):
            hip.set_option("spmv_relabel", relabel)
            a = DeviceCsMat.from_host(*a_host)
            xv = DeviceVec.from_host(x)
            y1 = (a * xv).to_host()
            y2 = (a * xv).to_host()                  # second SpMV: cached plan + scratch
            assert np.array_equal(y1, y2)
            yacc = DeviceVec.from_host(np.ones(n))
            prod.mul_acc_mat_vec_csr(a, xv, yacc)
            assert rel_err(yacc.to_host(), ref + 1.0) <= TOL
            out[relabel] = y1
            assert rel_err(y1, ref) <= TOL
        assert rel_err(out[1], out[2]) <= 1e-13 and not np.array_equal(out[1], np.zeros(n))
    finally:
        hip.set_option("spmv_xcs", 0)
        hip.set_option("spmv_relabel", 0)


def test_laplacian_componentwise_bound(hip):
    # config 3 shape at test size; mixed signs -> |dy_i| <= 1e-10 (|A||x|)_i  
    from oracle import oracle
    shape, ip, ix, dt = oracle.grid_laplacian(300, 300)
    from sprs_amd import gen
    x = gen.dense_vector(shape[0]).numpy()
    y = gpu_spmv(hip, shape, ip, ix, dt, x)
    ref = oracle_spmv(shape, ip, ix, dt, x)
    bound = oracle_spmv(shape, ip, ix, np.abs(dt), np.abs(x))
    assert np.all(np.abs(y - ref) <= TOL * bound)
    border = np.diff(ip.astype(np.int64)) == 1
    assert np.array_equal(y[border], x[border])          # Dirichlet rows are 1.0 * x exactly


RAGGED = [
    [0, 0, 0],
    [1],
    [0, 1, 0, 2, 0],
    [63, 64, 65, 0, 1],
    [4095, 1, 0, 4097, 2],
    [4096],
    [4096, 4096],
    [5000, 0, 0, 3],
    [10000, 1, 1, 1, 9000, 0, 0, 0, 0, 70, 64, 63, 200],
    [0] * 3000 + [7] + [0] * 3000,
    [1] * 9000,
    [2, 0] * 5000,
    [30000],
    [3] * 100 + [20000] + [5] * 100 + [0] * 50,
]


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("lens", RAGGED, ids=[str(i) for i in range(len(RAGGED))])
def test_ragged_rows(hip, lens, kernel):
    hip.set_option("spmv_kernel", kernel)
    try:
        cols = 40000
        shape, ip, ix, dt = ragged_csr(lens, cols, seed=len(lens))
        rng = np.random.default_rng(1)
        x = rng.random(cols) + 0.5
        for xcs, split, idx32, tile in ((2, 64, 1, 4096), (1, 64, 1, 4096), (1, 2, 0, 2048), (1, 5000, 1, 2048),
                                        (2, 64, 1, 2048)):
            hip.set_option("spmv_sort_tiles", 1 if split != 2 else 0)
            hip.set_option("spmv_xcs", xcs)
            hip.set_option("spmv_xcs_split", split)
            hip.set_option("spmv_xcs_idx32", idx32)
            hip.set_option("spmv_tile", tile)
            y = gpu_spmv(hip, shape, ip, ix, dt, x)
            ref = oracle_spmv(shape, ip, ix, dt, x)
            assert rel_err(y, ref) <= TOL
            y0 = rng.random(shape[0])
            y2 = gpu_spmv(hip, shape, ip, ix, dt, x, y=y0)
            assert rel_err(y2, oracle_spmv(shape, ip, ix, dt, x, y=y0)) <= TOL
            lens_a = np.array(lens)
            assert np.array_equal(y2[lens_a == 0], y0[lens_a == 0])
    finally:
        hip.set_option("spmv_kernel", 0)
        hip.set_option("spmv_xcs", 0)
        hip.set_option("spmv_xcs_split", 32)
        hip.set_option("spmv_xcs_idx32", 1)
        hip.set_option("spmv_tile", 0)
        hip.set_option("spmv_sort_tiles", 0)


def test_zero_sized(hip):
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd import prod
    z = np.zeros(0, dtype=np.uint64)
    a = DeviceCsMat.from_host((0, 7), np.zeros(1, dtype=np.uint64), z, np.zeros(0))
    y = a * DeviceVec.from_host(np.ones(7))
    assert y.n == 0
    a = DeviceCsMat.from_host((4, 0), np.zeros(5, dtype=np.uint64), z, np.zeros(0))
    y = (a * DeviceVec.from_host(np.zeros(0))).to_host()
    assert np.array_equal(y, np.zeros(4))
    yv = DeviceVec.from_host(np.arange(4.0))
    prod.mul_acc_mat_vec_csr(a, DeviceVec.from_host(np.zeros(0)), yv)
    assert np.array_equal(yv.to_host(), np.arange(4.0))


def test_contract_violations(hip, golden):
    from sprs_amd import SprsHipError, _ffi, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    shape, ip, ix, dt = as_csr(golden["mat3"])                 # 5 x 4
    a = DeviceCsMat.from_host(shape, ip, ix, dt)
    with pytest.raises(SprsHipError, match="Dimension mismatch") as e:   # prod.rs:114-117
        prod.mul_acc_mat_vec_csr(a, DeviceVec.zeros(5), DeviceVec.zeros(5))
    assert e.value.status == _ffi.DIM_MISMATCH
    with pytest.raises(SprsHipError, match="Dimension mismatch"):
        prod.mul_acc_mat_vec_csr(a, DeviceVec.zeros(4), DeviceVec.zeros(4))
    shape, ip, ix, dt = as_csr(golden["mat1_csc"])
    csc = DeviceCsMat.from_host(shape, ip, ix, dt, storage=_ffi.CSC)
    with pytest.raises(SprsHipError, match="Storage mismatch") as e:     # prod.rs:118
        prod.mul_acc_mat_vec_csr(csc, DeviceVec.zeros(5), DeviceVec.zeros(5))
    assert e.value.status == _ffi.STORAGE_MISMATCH
    # CsMat::new validation (sparse.rs:300-358)
    with pytest.raises(SprsHipError) as e:
        DeviceCsMat.from_host((2, 3), np.array([0, 2, 3], dtype=np.uint64),
                              np.array([2, 1, 0], dtype=np.uint64), np.ones(3))
    assert e.value.status == _ffi.BAD_STRUCTURE and "not sorted" in str(e.value)
    with pytest.raises(SprsHipError) as e:
        DeviceCsMat.from_host((2, 3), np.array([0, 2, 3], dtype=np.uint64),
                              np.array([0, 3, 0], dtype=np.uint64), np.ones(3))
    assert e.value.status == _ffi.BAD_STRUCTURE


def test_sliced_view_upload(hip, golden):
    # slice_outer views have a non-zero-based indptr (indptr.rs:118-124); upload rebases
    shape, ip, ix, dt = as_csr(golden["mat1"])
    x = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    full = gpu_spmv(hip, shape, ip, ix, dt, x)
    s = int(ip[1])
    part = gpu_spmv(hip, (3, 5), ip[1:5], ix[s:int(ip[4])], dt[s:int(ip[4])], x)
    assert np.array_equal(part, full[1:4])


def test_host_one_shot(hip, golden):
    from sprs_amd import _ffi
    fx = golden["mul_csr_vec"]
    shape, ip, ix, dt = as_csr(fx)
    x = np.array(fx["x"])
    y = np.array([1.0, 1.0, 1.0, 1.0, 1.0])
    vp = lambda a: C.c_void_p(a.ctypes.data)
    _ffi.check(_ffi.lib.sprs_hip_spmv_f64_host(5, 5, vp(ip), 8, vp(ix), 8, vp(dt), vp(x), 5, vp(y), 5, 1))
    assert np.all(np.abs(y - (1.0 + np.array(fx["expected"]))) < fx["epsilon"])


def test_run_to_run_deterministic(hip):
    from sprs_amd import gen
    n = 50000
    indptr, indices, data = gen.rmat_csr(n, 20, seed=11)
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    x = gen.dense_vector(n).numpy()
    for xcs in (1, 2):
        hip.set_option("spmv_xcs", xcs)
        try:
            ys = [gpu_spmv(hip, (n, n), ip, ix, dt, x) for _ in range(3)]
        finally:
            hip.set_option("spmv_xcs", 0)
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])


def test_stored_zeros_and_nonfinite(hip):
    # stored zeros are multiplied, not skipped: 0 * inf -> NaN propagates (prod.rs:120-126)
    ip = np.array([0, 2, 3], dtype=np.uint64)
    ix = np.array([0, 1, 1], dtype=np.uint64)
    dt = np.array([0.0, 2.0, 0.0])
    y = gpu_spmv(hip, (2, 2), ip, ix, dt, np.array([np.inf, 1.0]))
    assert np.isnan(y[0]) and y[1] == 0.0


def test_full_size_properties_config2(hip):
    """BASELINE config 2 size (R-MAT 1M x 1M, ~16 nnz/row) generated on the
    GPU; size-independent checks: linearity, row sums against an independent
    torch segment reduction, and agreement of the two kernels."""
    import torch
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    dev = torch.device("cuda", 0)
    n = 1_000_000
    indptr, indices, data = gen.rmat_csr(n, 16, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    x = gen.dense_vector(n, seed=3, device=dev)
    z = gen.dense_vector(n, seed=4, device=dev)

    def mul(v):
        out = torch.empty(n, dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(v), out=DeviceVec.borrow(out))
        torch.cuda.synchronize()
        return out

    ax, az = mul(x), mul(z)
    lin = mul(2.0 * x + 0.5 * z)
    ref = 2.0 * ax + 0.5 * az
    assert float(((lin - ref).abs() / ref.abs().clamp_min(1e-300)).max()) <= 1e-12 * 50
    ones = torch.ones(n, dtype=torch.float64, device=dev)
    rowsum = torch.zeros(n, dtype=torch.float64, device=dev)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), indptr[1:] - indptr[:-1])
    rowsum.index_add_(0, rows, data)
    got = mul(ones)
    assert float(((got - rowsum).abs() / rowsum.abs().clamp_min(1e-300)).max()) <= TOL
    for opt, val in (("spmv_kernel", 2), ("spmv_xcs", 2), ("spmv_xcs", 1)):
        hip.set_option(opt, val)
        try:
            other = mul(x)
        finally:
            hip.set_option(opt, 0)
        assert float(((other - ax).abs() / ax.abs().clamp_min(1e-300)).max()) <= TOL
    # and a 20k-row block against the oracle
    from oracle import oracle
    r0, r1 = 400000, 420000
    ip_h, ix_h, dt_h = a.slice_outer_to_host(r0, r1)
    yb = np.zeros(r1 - r0)
    oracle.mul_acc_mat_vec_csr((r1 - r0, n), ip_h, ix_h, dt_h, x.cpu().numpy(), yb)
    assert rel_err(ax[r0:r1].cpu().numpy(), yb) <= TOL


def test_cpp_host_mirror(hip):
    """include/sprs_hip.hpp (C++ host side, reference-shaped API) through the same C ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "host_mirror_test.bin")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


@pytest.mark.parametrize("parts", [2, 8])
def test_virtual_ranks_row_partition(hip, parts):
    """Multi-GPU path on ONE device: the nnz-balanced row blocks of sprs_amd.dist,
    each cut with slice_outer on the device and multiplied on its own handle, concatenate to the
    single-handle result bit for bit (per-row arithmetic does not depend on the partition when
    the same plan kind is used) and match the oracle."""
    import torch
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat, DeviceVec
    n = 80000
    indptr, indices, data = gen.rmat_csr(n, 12, seed=4)
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    x = gen.dense_vector(n).numpy()
    hip.set_option("spmv_xcs", 2)            # same (plain) plan for every shard and for the whole
    try:
        a = DeviceCsMat.from_host((n, n), ip, ix, dt)
        xv = DeviceVec.from_host(x)
        whole = (a * xv).to_host()
        cuts = gen.balanced_row_blocks(indptr, parts)
        y = np.zeros(n)
        for g in range(parts):
            blk = a.slice_outer(cuts[g], cuts[g + 1])
            assert blk.shape() == (cuts[g + 1] - cuts[g], n)
            y[cuts[g]:cuts[g + 1]] = (blk * xv).to_host()
    finally:
        hip.set_option("spmv_xcs", 0)
    assert rel_err(y, oracle_spmv((n, n), ip, ix, dt, x)) <= TOL
    assert rel_err(y, whole) <= 1e-14
    nnz_blocks = [int(ip[cuts[g + 1]] - ip[cuts[g]]) for g in range(parts)]
    assert max(nnz_blocks) <= ix.size / parts + int(np.diff(ip.astype(np.int64)).max())


def test_refresh_after_in_place_update(hip):
    """A wrapped handle caches plan copies of its arrays (XCD-sliced plan): after the caller scales the
    values in place, refresh() makes the next multiply see them."""
    import torch
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    dev = torch.device("cuda", 0)
    n = 60000
    indptr, indices, data = gen.rmat_csr(n, 16, device=dev)
    x = gen.dense_vector(n, device=dev)
    hip.set_option("spmv_xcs", 1)
    try:
        a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
        y1 = torch.empty(n, dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y1))
        torch.cuda.synchronize()
        data.mul_(2.0)
        a.refresh()
        y2 = torch.empty(n, dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y2))
        torch.cuda.synchronize()
    finally:
        hip.set_option("spmv_xcs", 0)
    assert torch.equal(y2, 2.0 * y1)          # scaling by 2 is exact in binary floating point


def _whole_vector_vs_oracle(indptr, indices, data, x, y, n, bound_componentwise=False):
    """whole GPU result against the oracle's serial SpMV (prod.rs:120-126) on the host copy of the same arrays"""
    from oracle import oracle
    ip_h, ix_h, dt_h = indptr.cpu().numpy().view(np.uint64), indices.cpu().numpy().view(np.uint64), data.cpu().numpy()
    x_h = x.cpu().numpy()
    ref = np.zeros(n)
    oracle.mul_acc_mat_vec_csr((n, n), ip_h, ix_h, dt_h, x_h, ref)
    got = y.cpu().numpy()
    if bound_componentwise:
        bound = np.zeros(n)
        oracle.mul_acc_mat_vec_csr((n, n), ip_h, ix_h, np.abs(dt_h), np.abs(x_h), bound)
        assert np.all(np.abs(got - ref) <= TOL * bound)
    else:
        assert rel_err(got, ref) <= TOL
    empty = np.diff(ip_h.astype(np.int64)) == 0
    assert np.all(got[empty] == 0.0)
    return ref


def test_config2_full_size_whole_vector(hip):
    """BASELINE config 2 (R-MAT 1M x 1M, ~16 nnz/row) at full size: EVERY component of y against the oracle
    (<= 1e-10 relative; positive data, no cancellation), for the plain plan and for the banded one."""
    import torch
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    dev = torch.device("cuda", 0)
    n = 1_000_000
    indptr, indices, data = gen.rmat_csr(n, 16, device=dev)
    x = gen.dense_vector(n, seed=3, device=dev)
    for band, kind in ((2, 1), (1, 3)):
        hip.set_option("spmv_band", band)
        try:
            a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
            y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
            prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y))
            torch.cuda.synchronize()
            assert a.spmv_plan_info()[0] == kind
        finally:
            hip.set_option("spmv_band", 0)
        _whole_vector_vs_oracle(indptr, indices, data, x, y, n)


def test_plan_policy_call_count_rule(hip):
    """a handle's first SpMV must not pay for a plan that copies the matrix.  Auto mode, a matrix the
    banded plan applies to: multiply 1 -> plain tile index (kind 1, a few hundred KB), multiply 2 -> the banded copy (kind 3);
    option spmv_plan_defer = 0 and sprs_hip_csmat_prepare build it at once; a forced plan (spmv_band = 1) is never deferred; the
    one-shot host entry never builds a copy; every result is the oracle's within the tolerance."""
    import torch
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    dev = torch.device("cuda", 0)
    n = 700_000
    indptr, indices, data = gen.rmat_csr(n, 24, device=dev)
    x = gen.dense_vector(n, seed=3, device=dev)

    def mul(a):
        y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y))
        torch.cuda.synchronize()
        return y

    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    assert a.spmv_plan_info() == (0, 0)
    y1 = mul(a)
    kind1, bytes1 = a.spmv_plan_info()
    assert kind1 == 1 and bytes1 < indices.numel() // 16           # the tile index: 8 bytes per 2 - 4 K entries
    y2 = mul(a)
    kind2, bytes2 = a.spmv_plan_info()
    assert kind2 == 3 and bytes2 > indices.numel() * 8              # the banded copy
    assert torch.equal(mul(a), y2)                                  # from here on the same plan, the same bits
    _whole_vector_vs_oracle(indptr, indices, data, x, y1, n)
    _whole_vector_vs_oracle(indptr, indices, data, x, y2, n)
    a.refresh()                                                     # values changed: the handle has multiplied before, the full plan comes back at once
    mul(a)
    assert a.spmv_plan_info()[0] == 3
    b = DeviceCsMat.wrap_torch((n, n), indptr, indices, data).prepare()
    assert b.spmv_plan_info()[0] == 3 and torch.equal(mul(b), y2)
    b.prepare()                                                     # idempotent
    assert b.spmv_plan_info()[0] == 3
    hip.set_option("spmv_plan_defer", 0)
    try:
        c = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
        assert torch.equal(mul(c), y2) and c.spmv_plan_info()[0] == 3
    finally:
        hip.set_option("spmv_plan_defer", 1)
    hip.set_option("spmv_band", 1)
    try:
        d = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
        mul(d)
        assert d.spmv_plan_info()[0] == 3
    finally:
        hip.set_option("spmv_band", 0)


def test_config3_full_size_whole_vector(hip):
    """BASELINE config 3 (5-point Laplacian of a 4096 x 4096 grid, heat.rs:45-80) at full size: every component
    within the componentwise bound |dy_i| <= 1e-10 (|A||x|)_i (mixed signs), Dirichlet rows exact."""
    import torch
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    dev = torch.device("cuda", 0)
    g = 4096
    n = g * g
    indptr, indices, data = gen.grid_laplacian(g, g, device=dev)
    x = gen.dense_vector(n, seed=3, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y))
    torch.cuda.synchronize()
    _whole_vector_vs_oracle(indptr, indices, data, x, y, n, bound_componentwise=True)
    border = (indptr[1:] - indptr[:-1]) == 1
    assert torch.equal(y[border], x[border])             # border rows are 1.0 * x exactly


def test_config4_full_size_whole_vector_and_determinism(hip):
    """BASELINE config 4 / the north star's target matrix (R-MAT 10M x 10M, ~32 nnz/row, usize indices) on one
    GPU with the DEFAULT plan (banded: hot columns from LDS): every component against the oracle, the
    accumulate form (prod.rs:103-127) against y0 + A x, and five SpMVs bit-identical."""
    import torch
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    dev = torch.device("cuda", 0)
    n = 10_000_000
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    x = gen.dense_vector(n, seed=3, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    # plan policy: the FIRST multiply of a handle runs on the plain tile index, the banded copy comes with the second one ...
    y_first = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y_first))
    torch.cuda.synchronize()
    assert a.spmv_plan_info()[0] == 1
    ys = []
    for _ in range(5):
        y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
        prod.csmat_mul_vec(a, DeviceVec.borrow(x), out=DeviceVec.borrow(y))
        torch.cuda.synchronize()
        ys.append(y)
    assert a.spmv_plan_info()[0] == 3
    # ... both within the tolerance of the oracle (the two plans group a row's products differently: equal to rounding)
    _whole_vector_vs_oracle(indptr, indices, data, x, y_first, n)
    # ... and sprs_hip_csmat_prepare builds it before the first multiply of a fresh handle
    b = DeviceCsMat.wrap_torch((n, n), indptr, indices, data).prepare()
    assert b.spmv_plan_info()[0] == 3
    yb = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    prod.csmat_mul_vec(b, DeviceVec.borrow(x), out=DeviceVec.borrow(yb))
    torch.cuda.synchronize()
    assert torch.equal(yb, ys[0])
    del b, yb, y_first
    for y in ys[1:]:
        assert torch.equal(y, ys[0])
    ref = _whole_vector_vs_oracle(indptr, indices, data, x, ys[0], n)
    y0 = gen.dense_vector(n, seed=9, device=dev)
    yacc = y0.clone()
    prod.mul_acc_mat_vec_csr(a, DeviceVec.borrow(x), DeviceVec.borrow(yacc))
    torch.cuda.synchronize()
    assert rel_err(yacc.cpu().numpy(), ref + y0.cpu().numpy()) <= TOL
    empty = (indptr[1:] - indptr[:-1]) == 0
    assert torch.equal(yacc[empty], y0[empty])            # empty rows untouched, bit for bit
