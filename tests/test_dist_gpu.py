"""The row-sharded SpMV inside the library (sprs_hip_dist_*, sprs_amd/csrc/dist.hip) on the device(s) this box has.
One GPU: world = 1 exercises the sub-block cut (slice_outer + rebased indptr, slicing.rs:65-89, indptr.rs:206-214) and
the multiply; with two or more GPUs the RCCL exchange runs too (one process per GPU, spawned here).  The CPU (gloo)
tests of the partition and exchange logic are tests/test_dist_cpu.py."""
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


@pytest.mark.parametrize("nsub", [1, 2, 5])
def test_world_of_one(hip, nsub):
    from oracle import oracle
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.dist import DistSpMV
    n = 30000
    indptr, indices, data = gen.rmat_csr(n, 12, seed=4)
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    a = DeviceCsMat.from_host((n, n), ip, ix, dt)
    d = DistSpMV((n, n), a, [0, n], rank=0, world=1, nsub=nsub)
    x = gen.dense_vector(n, seed=2).numpy()
    y = DeviceVec.from_host(np.full(n, np.nan))
    d.spmv(DeviceVec.from_host(x), y)
    ref = np.zeros(n)
    oracle.mul_acc_mat_vec_csr((n, n), ip, ix, dt, x, ref)
    assert rel_err(y.to_host(), ref) <= 1e-10
    with pytest.raises(hip.SprsHipError) as e:          # contract: row_starts must cover the rows, shapes must agree
        DistSpMV((n, n), a, [0, n - 1], rank=0, world=1)
    assert e.value.status == hip._ffi.INVALID_ARG
    with pytest.raises(hip.SprsHipError) as e:
        DistSpMV((n, n + 1), a, [0, n], rank=0, world=1)
    assert e.value.status == hip._ffi.DIM_MISMATCH


def test_world_of_one_through_rccl(hip):
    """what a 1-GPU box can run of the RCCL route: the library loads librccl itself, makes the id, initialises a communicator
    of one rank, runs its (empty) grouped exchanges on the second stream behind every sub-block and joins back"""
    from oracle import oracle
    from sprs_amd import gen
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.dist import DistSpMV
    n = 40000
    indptr, indices, data = gen.rmat_csr(n, 10, seed=5)
    ip, ix, dt = indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy()
    a = DeviceCsMat.from_host((n, n), ip, ix, dt)
    uid = DistSpMV.unique_id()
    assert len(uid) == 128 and any(uid)
    d = DistSpMV((n, n), a, [0, n], rank=0, world=1, unique_id=uid, nsub=3)
    x = gen.dense_vector(n, seed=6).numpy()
    ref = np.zeros(n)
    oracle.mul_acc_mat_vec_csr((n, n), ip, ix, dt, x, ref)
    xv = DeviceVec.from_host(x)
    for _ in range(3):                                  # the events / second stream are reused call after call
        y = DeviceVec.from_host(np.full(n, np.nan))
        d.spmv(xv, y)
        assert rel_err(y.to_host(), ref) <= 1e-10
    del d


def test_row_sharded_bicgstab_on_the_hip_kernels(hip):
    """the row-sharded BiCGSTAB driver (sprs_amd/dist.py, twin of linalg/bicgstab.rs:117-229) with the HIP SpMV as its local
    kernel — a world of one: every SpMV of the solver goes through the device path; iterates equal the serial oracle's to
    rounding"""
    import torch
    from oracle import oracle
    from sprs_amd import gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.dist import RowShardedBiCGSTAB, RowShardedSpMV
    dev = torch.device("cuda", 0)
    grid = 96
    n = grid * grid
    indptr, indices, data = gen.grid_laplacian(grid, grid, device=dev)
    data = data.clone()
    rows_of = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]))
    diag = indices == rows_of
    data[diag] = data[diag].abs() + 1.5
    data[indices > rows_of] *= 0.7
    handles = {}

    def local_spmv(block, xv, y_block):
        if id(block) not in handles:
            rows_b, cols_b, ip, ix, dt = block
            handles[id(block)] = DeviceCsMat.wrap_torch((rows_b, cols_b), ip, ix, dt)
        prod.csmat_mul_vec(handles[id(block)], DeviceVec.borrow(xv), out=DeviceVec.borrow(y_block))
        torch.cuda.synchronize()

    sh = RowShardedSpMV((n, n), indptr, indices, data, local_spmv)
    b = gen.dense_vector(n, seed=5, device=dev)
    x0 = torch.zeros(n, dtype=torch.float64, device=dev)
    sol = RowShardedBiCGSTAB.solve(sh, x0, b, 1e-10, 600)
    x_ref, info = oracle.bicgstab((n, n), indptr.cpu().numpy().astype(np.uint64), indices.cpu().numpy().astype(np.uint64),
                                  data.cpu().numpy(), x0.cpu().numpy(), b.cpu().numpy(), 1e-10, 600)
    assert sol.converged and info["converged"]
    assert abs(sol.iteration_count - info["iteration_count"]) <= 2
    assert rel_err(sol.x_full().cpu().numpy(), x_ref) <= 1e-8


def _rank_main(rank, world, port, n, out):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    import sprs_amd
    from sprs_amd import _ffi, gen
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.dist import DistSpMV
    _ffi.check(_ffi.lib.sprs_hip_set_device(rank))
    indptr, indices, data = gen.rmat_csr(n, 16, seed=7, device=dev)
    cuts = gen.balanced_row_blocks(indptr, world)
    full = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    block = full.slice_outer(cuts[rank], cuts[rank + 1])
    d = DistSpMV((n, n), block, cuts, rank, world, unique_id=DistSpMV.broadcast_id(dev), nsub=3)
    x = gen.dense_vector(n, seed=3, device=dev)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    for _ in range(3):                                   # y fed back would be the next x: here just repeated
        d.spmv(DeviceVec.borrow(x), DeviceVec.borrow(y), stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref = torch.empty(n, dtype=torch.float64, device=dev)
    from sprs_amd import prod
    prod.csmat_mul_vec(full, DeviceVec.borrow(x), out=DeviceVec.borrow(ref))
    torch.cuda.synchronize()
    ok = bool(((y - ref).abs() <= 1e-10 * ref.abs()).all())
    # the same handle on the peer-store route: the same multiply kernels, rows moved by stores instead of ncclSend / ncclRecv
    y2 = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    d.connect_peers(dev).set_route("peer")
    for _ in range(3):
        d.spmv(DeviceVec.borrow(x), DeviceVec.borrow(y2), stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(y, y2))
    d.set_route("rccl")
    dist.barrier()
    if rank == 0:
        open(out, "w").write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def _peer_rank_main(rank, world, port, n, out, devices):
    """one rank of the PEER-route test: gloo is only the bootstrap (window handles, barriers); the rows travel by stores into
    the peers' receive windows (sprs_amd/csrc/dist.hip) — through IPC mappings of the same device when the ranks share one"""
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    di = devices[rank]
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    dist.init_process_group("gloo")
    import sprs_amd
    from sprs_amd import _ffi, gen, prod
    from sprs_amd.device import DeviceCsMat, DeviceVec
    from sprs_amd.dist import DistSpMV
    _ffi.check(_ffi.lib.sprs_hip_set_device(di))
    indptr, indices, data = gen.rmat_csr(n, 16, seed=7, device=dev)
    cuts = gen.balanced_row_blocks(indptr, world, row_weight=8.0)
    full = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    block = full.slice_outer(cuts[rank], cuts[rank + 1])
    d = DistSpMV((n, n), block, cuts, rank, world, unique_id=None, nsub=3)
    ok = True
    try:                                                 # no RCCL communicator, route not connected yet: the multiply must refuse
        d.spmv(DeviceVec.borrow(torch.zeros(n, dtype=torch.float64, device=dev)), DeviceVec.borrow(torch.zeros(n, dtype=torch.float64, device=dev)))
        ok = False
    except sprs_amd.SprsHipError as e:
        ok = ok and e.status == _ffi.INVALID_ARG
    d.connect_peers(dev).set_route("peer")
    ok = ok and d.route() == "peer"
    x = gen.dense_vector(n, seed=3, device=dev)
    ref = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    for it in range(5):                                  # y is fed back as the next x: both copies of the window are in use in turn
        y.fill_(float("nan"))
        d.spmv(DeviceVec.borrow(x), DeviceVec.borrow(y), stream=torch.cuda.current_stream())
        prod.csmat_mul_vec(full, DeviceVec.borrow(x), out=DeviceVec.borrow(ref))
        torch.cuda.synchronize()
        # the own block is the same kernel on a row block (another plan than the whole matrix: equal to rounding); every rank must hold ALL rows
        ok = ok and bool(((y - ref).abs() <= 1e-10 * ref.abs()).all())
        x = y / y.abs().max()
    # every rank has the same gathered vector, bit for bit
    chk = torch.tensor([float((y * (1.0 + (torch.arange(n, device=dev) % 7))).sum().item())], dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ok = ok and float(lo.item()) == float(hi.item())
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()                                       # nobody frees its window while a peer may still store into it
    del d
    if rank == 0:
        open(out, "w").write("ok" if float(flag.item()) > 0.5 else "mismatch")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_peer_route_ranks_sharing_this_gpu(hip, tmp_path, world):
    """the second exchange route as far as ONE GPU can run it: `world` processes on device 0, every
    rank's receive window exported as a HIP IPC handle and mapped by the others, five chained SpMVs whose sub-blocks are pushed
    into the peers' windows while the next sub-block multiplies; every rank ends with the whole vector, equal to the one-handle
    product and identical across ranks.  (With two devices test_two_gpus_rccl_exchange also compares this route with RCCL's.)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_peer_rank_main, args=(world, 29541 + world, 300000, out, [0] * world), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_two_gpus_rccl_exchange(hip, tmp_path):
    """needs two devices: one process per GPU, the id broadcast through torch.distributed, three SpMVs with the
    exchange overlapping the sub-block multiplies; every rank's gathered y equals the single-GPU result"""
    if hip.device_count() < 2:
        pytest.skip("one GPU on this box: the RCCL exchange needs two (hipGetDeviceCount() >= 2)")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_rank_main, args=(2, 29533, 400000, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _run_bench(extra, timeout=600):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, cwd=root, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected from rank 0: %r" % r.stdout[-1500:]
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_bench_n_ranks_on_this_gpu(hip, world):
    """the N > 1 path of bench.py must be executable where only ONE GPU exists, so that the first real
    8-GPU run cannot die in Python.  `python bench.py --gpus N --backend gloo --devices 0,0[,0]` goes through everything the
    driver's command goes through except RCCL itself: the self-launch under torch.distributed.run, the cost-balanced partition
    (slicing.rs:65-89, indptr.rs:206-214), RowShardedSpMV on the HIP kernels of THIS library (banded plan per block), the route
    agreement, the timed loop, the secondary timings and every JSON key — the y blocks travel through host memory instead of
    xGMI.  The distributed result is checked against the oracle block by block inside the run (`parity`)."""
    out, err = _run_bench(["--gpus", str(world), "--backend", "gloo", "--devices", ",".join(["0"] * world),
                           "--workload", "rmat:400000:24", "--steps", "4", "--warmup", "1"])
    assert out["n_gpus"] == world and out["steps"] == 4 and out["warmup"] == 1
    assert out["metric"] == "CSR SpMV GFLOP/s" and out["unit"] == "GFLOP/s" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["scaling"] == "strong" and out["higher_is_better"] is True and out["dtype"] == "f64"
    assert ("x%d" % world) in out["config"]["partition"] and "gloo" in out["config"]["partition"]
    ex = out["exchange"]
    assert ex["timed_route"] == "torch" and ex["backend"] == "gloo" and ex["devices"] == ",".join(["0"] * world)
    assert isinstance(ex["multiply_only_ms"], float) and isinstance(ex["torch_route_ms"], float) and ex["lib_route_ms"] is None
    # the peer-store route needs no RCCL: it runs here too (windows mapped through IPC on the one device) and must agree with the torch route
    assert isinstance(ex["peer_route_ms"], float) and ex["routes_agree"]["ok"] and ex["routes_agree"]["peer_vs_torch_max_rel_diff"] <= 1e-10
    assert ex["torch_route_ms"] >= ex["multiply_only_ms"] * 0.5
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["algorithmic_bytes_per_launch"] > 0
    assert rf["plan"] in ("banded copy (hot columns from LDS)", "nnz tiles", "xcd-sliced copy")
    par = out["parity"]
    assert par["ok"] and par["gathered_y_identical"] and par["max_rel_err_vs_oracle"] <= 1e-10
    assert "cpu_baseline" not in out                     # N = 1 only (the contract)
    assert "not attempted" in err                        # the line says that the RCCL route was not taken


def test_bench_single_rank_keys(hip):
    """the N = 1 line of the driver's contract on a small workload: roofline + cpu_baseline + parity objects present"""
    out, _ = _run_bench(["--workload", "rmat:300000:16", "--steps", "5", "--warmup", "2"])
    assert out["n_gpus"] == 1 and out["config"]["partition"] == "single GPU"
    for key in ("roofline", "cpu_baseline", "parity"):
        assert key in out
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1 and out["parity"]["ok"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"])
