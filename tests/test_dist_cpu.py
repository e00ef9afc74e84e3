"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups running the
row-sharded SpMV driver (sprs_amd/dist.py) — partition, block rebasing and the
direct all-gather-v exchange.  The local multiply is injected (the oracle
stands in for the HIP kernel, which needs a GPU); on the GPU box bench.py wires
the same driver to the HIP path over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, steps, ret, exchange="direct"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from sprs_amd import gen
        from sprs_amd.dist import RowShardedSpMV
        indptr, indices, data = gen.rmat_csr(n, 8, seed=3)

        def local_spmv(block, x, y_block):           # CPU stand-in for the HIP kernel
            rows, cols, ip, ix, dt = block
            assert int(ip[0]) == 0                   # rebased (to_proper)
            y = np.zeros(rows)
            oracle.mul_acc_mat_vec_csr((rows, cols), ip.numpy().astype(np.uint64), ix.numpy().astype(np.uint64),
                                       dt.numpy(), x.numpy(), y)
            y_block.copy_(torch.from_numpy(y))

        sh = RowShardedSpMV((n, n), indptr, indices, data, local_spmv, exchange=exchange)
        assert sh.world == world and sh.rank == rank
        x = gen.dense_vector(n)
        for _ in range(steps):                       # y becomes the next x, as an iterative solver would
            y = sh.step(x).clone()
            x = y / float(y.abs().max())
        full = np.zeros(n)
        x_ref = gen.dense_vector(n).numpy()
        for _ in range(steps):
            full[:] = 0
            oracle.mul_acc_mat_vec_csr((n, n), indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64),
                                       data.numpy(), x_ref, full)
            y_ref = full.copy()
            x_ref = y_ref / np.abs(y_ref).max()
        ok = np.array_equal(y.numpy(), y_ref)
        blocks = [int(indptr[b] - indptr[a]) for a, b in zip(sh.cuts, sh.cuts[1:])]
        ret[rank] = (bool(ok), sh.block_nnz, blocks, sh.cuts)
    finally:
        dist.destroy_process_group()


def test_row_sharded_spmv_gloo_allgather_fallback():
    """the padded all_gather the driver falls back to when the grouped send/recv is refused"""
    n = 3000
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), n, 2, ret, "allgather"), nprocs=2, join=True)
    assert len(ret) == 2 and all(ret[r][0] for r in range(2))


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_spmv_gloo_host_staged_exchange(world):
    """the exchange a GPU run takes under gloo (bench.py --backend gloo: several ranks on ONE GPU): own block -> host buffer,
    grouped send / recv of host slices, peers' blocks copied into y — here with y itself on the host"""
    n = 5000
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), n, 2, ret, "staged"), nprocs=world, join=True)
    assert len(ret) == world and all(ret[r][0] for r in range(world))


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_spmv_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    n = 20000
    mp.spawn(_worker, args=(world, _free_port(), n, 2, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        ok, block_nnz, blocks, cuts = ret[rank]
        assert ok, "rank %d: gathered y differs from the serial oracle" % rank
        assert block_nnz == blocks[rank]
    total = sum(ret[0][2])
    assert max(ret[0][2]) <= total / world * 1.6       # cost-balanced (nnz + 8/row) despite the power law


def _spgemm_worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from sprs_amd import gen
        from sprs_amd.dist import RowShardedSpGEMM
        indptr, indices, data = gen.rmat_csr(n, 6, seed=9)
        u = lambda t: t.numpy().astype(np.uint64)

        def local_spgemm(a_block, b):                 # CPU stand-in for the HIP SpGEMM
            (ar, ac), aip, aix, adt = a_block
            (br, bc), bip, bix, bdt = b
            assert int(aip[0]) == 0
            return oracle.mul_csr_csr((ar, ac), u(aip), u(aix), adt.numpy(), (br, bc), u(bip), u(bix), bdt.numpy(),
                                      threads=1)

        sh = RowShardedSpGEMM(((n, n), indptr, indices, data), ((n, n), indptr, indices, data), local_spgemm)
        shape, cip, cix, cdt = sh.multiply()
        assert shape == (sh.r1 - sh.r0, n)
        full_ip = sh.gather_indptr(torch.from_numpy(cip.astype(np.int64)))
        rshape, rip, rix, rdt = oracle.mul_csr_csr((n, n), u(indptr), u(indices), data.numpy(), (n, n), u(indptr),
                                                   u(indices), data.numpy(), threads=1)
        lo, hi = int(rip[sh.r0]), int(rip[sh.r1])
        ok = (np.array_equal(full_ip.numpy().astype(np.uint64), rip) and np.array_equal(cix, rix[lo:hi])
              and np.array_equal(cdt, rdt[lo:hi]))
        ret[rank] = (bool(ok), sh.block_products, sh.cuts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_spgemm_gloo(world):
    """SpGEMM by A-row blocks, B replicated, no data-path collective: every rank's block of C equals the
    same rows of the single-process product bit for bit, the gathered indptr equals the global one, and
    the blocks are balanced by products despite the power law."""
    ret = mp.Manager().dict()
    n = 4000
    mp.spawn(_spgemm_worker, args=(world, _free_port(), n, ret), nprocs=world, join=True)
    assert len(ret) == world and all(ret[r][0] for r in range(world))
    prods = [ret[r][1] for r in range(world)]
    assert max(prods) <= sum(prods) / world * 1.5


def _bicgstab_worker(rank, world, port, grid, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from sprs_amd import gen
        from sprs_amd.dist import RowShardedBiCGSTAB, RowShardedSpMV
        # the heat example's operator (sprs/examples/heat.rs:45-80): 5-point Laplacian with Dirichlet border rows,
        # made non-symmetric and diagonally dominant so that BiCGSTAB has something to do
        indptr, indices, data = gen.grid_laplacian(grid, grid)
        n = grid * grid
        data = data.clone()
        rows_of = torch.repeat_interleave(torch.arange(n), (indptr[1:] - indptr[:-1]))
        diag = indices == rows_of
        data[diag] = data[diag].abs() + 1.5
        data[indices > rows_of] *= 0.7
        ip, ix, dt = (indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy())

        def local_spmv(block, x, y_block):           # CPU stand-in for the HIP kernel
            rows, cols, bip, bix, bdt = block
            y = np.zeros(rows)
            oracle.mul_acc_mat_vec_csr((rows, cols), bip.numpy().astype(np.uint64), bix.numpy().astype(np.uint64),
                                       bdt.numpy(), x.numpy(), y)
            y_block.copy_(torch.from_numpy(y))

        sh = RowShardedSpMV((n, n), indptr, indices, data, local_spmv)
        b = gen.dense_vector(n, seed=5)
        x0 = torch.zeros(n, dtype=torch.float64)
        sol = RowShardedBiCGSTAB.solve(sh, x0, b, 1e-10, 400)
        x = sol.x_full().numpy()
        x_ref, info = oracle.bicgstab((n, n), ip, ix, dt, x0.numpy(), b.numpy(), 1e-10, 400)
        res = np.zeros(n)
        oracle.mul_acc_mat_vec_csr((n, n), ip, ix, dt, x, res)
        ret[rank] = (bool(sol.converged), int(sol.iteration_count), int(info["iteration_count"]), bool(info["converged"]),
                     float(np.max(np.abs(x - x_ref)) / np.max(np.abs(x_ref))), float(np.linalg.norm(res - b.numpy())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_bicgstab_gloo(world):
    """sprs::linalg::bicgstab over the row-sharded SpMV (every iteration gathers its operand): same control flow as the
    serial oracle (linalg/bicgstab.rs:117-229), iterates equal to rounding — same iteration count, solution within 1e-9,
    true residual below the tolerance"""
    ret = mp.Manager().dict()
    mp.spawn(_bicgstab_worker, args=(world, _free_port(), 40, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        conv, it, it_ref, conv_ref, dx, res = ret[rank]
        assert conv and conv_ref
        assert abs(it - it_ref) <= 2, (it, it_ref)
        assert dx <= 1e-9 and res <= 1e-8


def test_row_sharded_bicgstab_breakdown_like_the_reference():
    """A = I, b = x0 + e: the first step solves the system exactly and the recurrences divide 0 by 0.  The reference computes in
    IEEE floats (bicgstab.rs:193-223: omega = NaN, the iteration goes on and ends in Err / Ok through solve()); the row-sharded
    solver must not raise ZeroDivisionError there (world of one: no process group needed)."""
    from oracle import oracle
    from sprs_amd.dist import RowShardedBiCGSTAB, RowShardedSpMV
    n = 50
    indptr = torch.arange(n + 1, dtype=torch.int64)
    indices = torch.arange(n, dtype=torch.int64)
    data = torch.ones(n, dtype=torch.float64)

    def local_spmv(block, x, y_block):
        rows, cols, bip, bix, bdt = block
        y = np.zeros(rows)
        oracle.mul_acc_mat_vec_csr((rows, cols), bip.numpy().astype(np.uint64), bix.numpy().astype(np.uint64), bdt.numpy(), x.numpy(), y)
        y_block.copy_(torch.from_numpy(y))

    sh = RowShardedSpMV((n, n), indptr, indices, data, local_spmv)
    b = torch.linspace(1.0, 2.0, n, dtype=torch.float64)
    for x0 in (b.clone(), torch.zeros(n, dtype=torch.float64)):       # x0 already the solution (r = 0), and one exact step away
        sol = RowShardedBiCGSTAB.solve(sh, x0, b, 1e-12, 5)
        _, info = oracle.bicgstab((n, n), indptr.numpy().astype(np.uint64), indices.numpy().astype(np.uint64), data.numpy(),
                                  x0.numpy(), b.numpy(), 1e-12, 5)
        assert bool(sol.converged) == bool(info["converged"])
        assert int(sol.iteration_count) == int(info["iteration_count"])
