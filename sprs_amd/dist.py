"""Row-sharded SpMV across the GPUs of one node.

The reference has no distributed code; the shard is its own `slice_outer`
(sprs/src/sparse/slicing.rs:65-89): rank g owns the contiguous row block
[r_g, r_{g+1}) chosen so that every block has ~1/G of the cost (nnz + 8 per row), keeps a
full replica of x and computes y[r_g:r_{g+1}) = A[r_g:r_{g+1}, :] * x with the
single-GPU kernel.  The one exchange step is an all-gather-v of y.

MI355X-native choice for the exchange: xGMI is a full point-to-point mesh
(7 links per GPU), so a ring all-gather is bound by ONE link while a direct
exchange — every rank sends its block to its 7 peers at once — uses all seven.
The exchange is therefore issued as one grouped batch of send/recv pairs
(`batch_isend_irecv` == ncclGroupStart / ncclSend+ncclRecv / ncclGroupEnd on
the RCCL backend).  The same code runs on gloo for the CPU tests.

torch / torch.distributed are plumbing here (device memory, process group);
the multiply itself is `local_spmv`, by default the HIP path.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import gen


class RowShardedSpMV:
    """y = A * x with A split by nnz-balanced row blocks over the process group.

    indptr/indices/data: the FULL matrix as torch tensors on this rank's device
    (every rank generates or loads the same matrix; only the own block is
    kept).  local_spmv(block, x, y_block) multiplies the local block, where
    block = (rows, cols, indptr, indices, data) with a zero-based indptr.
    """

    def __init__(self, shape, indptr, indices, data, local_spmv, group=None, row_weight=None, exchange="direct"):
        assert exchange in ("direct", "allgather", "staged")   # staged: the direct exchange through host memory (what a GPU run under gloo takes)
        self.mode = exchange
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows, self.cols = shape
        # blocks of equal cost nnz + row_weight * rows.  A row costs ~5 entries of compute on MI355X; a larger weight evens out the y
        # blocks of the exchange (R-MAT keeps its long rows in front: the last blocks have the most rows) at the price of uneven
        # multiplies.  Chosen on the one-GPU model max(block multiply) + max(y block) / one xGMI link (scripts/virtual_ranks.py,
        # profiles/r13t_virtual_ranks_row_weight_sweep.jsonl): 8 is best up to 4 ranks (0.562 ms against 0.582 at 16), 16 - 24 at 8
        # ranks (0.358 against 0.378 at 8) — a model, not a multi-GPU measurement.  (Round 5's default depended on the world size, 8 up to
        # 4 ranks and 16 above; see below.)
        # Round 6 (profiles/r16a_virtual_ranks_scaling_model.jsonl): with the small-plan kernels 8 is best at 8 ranks as well (modelled step
        # 0.3545 against 0.3625 ms at 16) — one default for every world size again.
        if row_weight is None:
            row_weight = 8.0
        self.row_weight = row_weight
        self.cuts = gen.balanced_row_blocks(indptr, self.world, row_weight=row_weight)
        r0, r1 = self.cuts[self.rank], self.cuts[self.rank + 1]
        lo, hi = int(indptr[r0]), int(indptr[r1])
        # slice_outer + to_proper (indptr.rs:206-214): rebase the block's indptr
        # (clone: fresh, 16-byte aligned allocations the full arrays can be freed behind)
        self.block = (r1 - r0, self.cols, (indptr[r0:r1 + 1] - indptr[r0]).contiguous(),
                      indices[lo:hi].clone(), data[lo:hi].clone())
        self.block_nnz = hi - lo
        self.total_nnz = int(indptr[-1]) - int(indptr[0])
        self.r0, self.r1 = r0, r1
        self.local_spmv = local_spmv
        self.y = torch.zeros(self.rows, dtype=torch.float64, device=indptr.device)

    def _exchange_allgather(self):
        """Fallback: one padded all_gather (ring or tree inside the library) + unpack.  Moves
        world * max_block doubles instead of rows, and a ring is bound by one xGMI link."""
        pad = max(b - a for a, b in zip(self.cuts, self.cuts[1:]))
        if not hasattr(self, "_ag"):
            self._ag = torch.zeros(self.world * pad, dtype=torch.float64, device=self.y.device)
            self._mine = torch.zeros(pad, dtype=torch.float64, device=self.y.device)
        self._mine[:self.r1 - self.r0].copy_(self.y[self.r0:self.r1])
        dist.all_gather_into_tensor(self._ag, self._mine, group=self.group)
        for peer in range(self.world):
            if peer != self.rank:
                a, b = self.cuts[peer], self.cuts[peer + 1]
                self.y[a:b].copy_(self._ag[peer * pad:peer * pad + (b - a)])

    def exchange(self):
        """all-gather-v of y: direct send/recv with every peer, one group."""
        if self.world == 1:
            return
        if self.mode == "allgather":
            return self._exchange_allgather()
        if self.mode == "staged":
            return self._exchange_staged()
        try:
            self._exchange_direct()
        except RuntimeError as e:                     # e.g. a backend without grouped P2P on this topology
            import sys
            print("sprs_amd.dist: direct exchange failed (%s); falling back to all_gather" % str(e).splitlines()[0],
                  file=sys.stderr)
            self.mode = "allgather"
            self._exchange_allgather()

    def _exchange_staged(self):
        """The same direct exchange for a backend without device-side P2P (gloo with the vectors on a GPU): the own block goes
        to pinned host memory once, the peers' blocks arrive there and are copied into y.  For running the N > 1 path where
        RCCL cannot come up (several ranks on one GPU); never the measured route of a real multi-GPU run."""
        if not hasattr(self, "_host"):
            self._host = torch.empty(self.rows, dtype=torch.float64)
            if self.y.is_cuda:
                self._host = self._host.pin_memory()
        h = self._host
        if self.y.is_cuda:
            # the previous exchange ended with non-blocking H2D copies out of `h`: they must be done before gloo writes into it
            # again (the blocking D2H copy below is no such guarantee: it is skipped for an empty block and orders only the
            # stream it runs on)
            ev = getattr(self, "_h2d_done", None)
            if ev is not None:
                ev.synchronize()
            torch.cuda.current_stream(self.y.device).synchronize()      # ... and the multiply that wrote the own block
        h[self.r0:self.r1].copy_(self.y[self.r0:self.r1])
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            a, b = self.cuts[peer], self.cuts[peer + 1]
            if self.r1 > self.r0:
                ops.append(dist.P2POp(dist.isend, h[self.r0:self.r1], peer, self.group))
            if b > a:
                ops.append(dist.P2POp(dist.irecv, h[a:b], peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for peer in range(self.world):
            if peer != self.rank:
                a, b = self.cuts[peer], self.cuts[peer + 1]
                self.y[a:b].copy_(h[a:b], non_blocking=True)
        if self.y.is_cuda:
            if getattr(self, "_h2d_done", None) is None:
                self._h2d_done = torch.cuda.Event()
            self._h2d_done.record(torch.cuda.current_stream(self.y.device))

    def _exchange_direct(self):
        if self.y.is_cuda and dist.get_backend(self.group) == "gloo":
            return self._exchange_staged()
        mine = self.y[self.r0:self.r1]
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            theirs = self.y[self.cuts[peer]:self.cuts[peer + 1]]
            if mine.numel():
                ops.append(dist.P2POp(dist.isend, mine, peer, self.group))
            if theirs.numel():
                ops.append(dist.P2POp(dist.irecv, theirs, peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def step(self, x):
        """One SpMV: local block multiply, then the exchange.  Returns the full y."""
        self.local_spmv(self.block, x, self.y[self.r0:self.r1])
        self.exchange()
        return self.y


class DistSpMV:
    """The row-sharded SpMV INSIDE the library (sprs_hip_dist_create / _spmv_f64 / _free, sprs_amd/csrc/dist.hip): the
    local block cut into sub-blocks, the direct RCCL exchange of a finished sub-block overlapping the multiply of the
    next.  The 128-byte RCCL id is made on one rank (`unique_id`) and handed to the others by whatever the host
    program has — here a torch.distributed broadcast (`broadcast_id`)."""

    def __init__(self, shape, local_block, row_starts, rank, world, unique_id=None, nsub=2):
        import ctypes as C
        import numpy as np
        from ._ffi import check, lib
        self.rows, self.cols = int(shape[0]), int(shape[1])
        self.block = local_block                      # keeps the handle alive (the library slices its own copies)
        rs = np.ascontiguousarray(row_starts, dtype=np.uint64)
        h = C.c_void_p()
        idbuf = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        check(lib.sprs_hip_dist_create(C.byref(h), idbuf, world, rank, self.rows, self.cols,
                                       rs.ctypes.data_as(C.POINTER(C.c_uint64)), local_block._h, nsub))
        self._h = h

    @staticmethod
    def unique_id():
        import ctypes as C
        from ._ffi import check, lib
        buf = (C.c_char * 128)()
        check(lib.sprs_hip_dist_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def broadcast_id(device, group=None, src=0):
        """rank `src` makes the id, everybody gets it (collective)"""
        buf = torch.zeros(128, dtype=torch.uint8, device=device)
        if dist.get_rank(group) == src:
            buf.copy_(torch.frombuffer(bytearray(DistSpMV.unique_id()), dtype=torch.uint8))
        dist.broadcast(buf, src=src, group=group)
        return bytes(buf.cpu().numpy().tobytes())

    # ---- the second exchange route: stores into the peers' receive windows over xGMI (sprs_hip.h, dist.hip) ------------
    def peer_handle(self):
        """this rank's receive window as a 64-byte HIP IPC handle"""
        import ctypes as C
        from ._ffi import check, lib
        buf = (C.c_char * 64)()
        check(lib.sprs_hip_dist_peer_handle(self._h, buf))
        return bytes(buf)

    def connect_peers(self, device, group=None):
        """collective: all-gather the 64-byte window handles through torch.distributed (whatever backend the group has) and
        map every peer's window; afterwards `set_route("peer")` is allowed.  A failure on ANY rank (no IPC for this kind of
        memory, a peer that cannot be mapped) is agreed on by all ranks before anybody raises: no rank is left waiting in a
        collective the failed one never enters."""
        import ctypes as C
        from ._ffi import check, lib
        world = dist.get_world_size(group)
        on = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        err = None
        try:
            handle = self.peer_handle()
        except Exception as e:                          # (still take part in the collectives below)
            err, handle = e, bytes(64)
        mine = torch.frombuffer(bytearray(handle), dtype=torch.uint8).to(on)
        got = [torch.zeros(64, dtype=torch.uint8, device=on) for _ in range(world)]
        dist.all_gather(got, mine, group=group)
        ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64, device=on)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if float(ok.item()) > 0.5:                      # every rank exported a window: map them
            blob = b"".join(bytes(t.cpu().numpy().tobytes()) for t in got)
            try:
                check(lib.sprs_hip_dist_peer_connect(self._h, (C.c_char * len(blob)).from_buffer_copy(blob), world))
            except Exception as e:
                err = e
            ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64, device=on)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)   # (also the barrier: every rank has mapped every window before anybody stores)
        if float(ok.item()) < 0.5:
            raise err if err is not None else RuntimeError("peer route: another rank could not export or map a receive window")
        return self

    def set_route(self, route):
        """"rccl" (grouped ncclSend / ncclRecv) or "peer" (stores into the peers' windows)"""
        from ._ffi import check, lib
        check(lib.sprs_hip_dist_set_route(self._h, {"rccl": 0, "peer": 1}[route]))
        return self

    def route(self):
        import ctypes as C
        from ._ffi import check, lib
        r = C.c_int32(0)
        check(lib.sprs_hip_dist_route(self._h, C.byref(r)))
        return ("rccl", "peer")[r.value]

    def comm_count(self):
        """ranks of the RCCL communicator as RCCL itself counts them (ncclCommCount; 1 for a world of one)"""
        import ctypes as C
        from ._ffi import check, lib
        n = C.c_int32(0)
        check(lib.sprs_hip_dist_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def spmv(self, x, y, stream=None):
        """y = A * x (collective); x, y: DeviceVec of length cols / rows on this rank's device"""
        import ctypes as C
        from ._ffi import check, lib
        from .prod import _stream_ptr
        check(lib.sprs_hip_dist_spmv_f64(self._h, C.c_void_p(x.ptr), x.n, C.c_void_p(y.ptr), y.n, _stream_ptr(stream)))
        return y

    def __del__(self):
        from ._ffi import lib
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:
            lib.sprs_hip_dist_free(h)
            self._h = None


class RowShardedBiCGSTAB:
    """`sprs::linalg::bicgstab::BiCGSTAB` (sprs/src/sparse/linalg/bicgstab.rs:117-229) over the row-sharded SpMV: the
    reason the all-gather of y exists — the gathered product of one iteration feeds the next.  Every rank owns the rows
    [r_g, r_{g+1}) of A and the same block of every solver vector (x, r, rhat, p, v, s, t, b); an SpMV gathers the full
    operand first (the driver's direct all-gather-v), multiplies the local row block, and leaves the result sharded; the
    four dot products of an iteration are local sums + one all_reduce each.  Control flow, restarts and the order of the
    scalar operations are the reference's (`step`, `soft_restart`, `hard_restart`, `solve`); the dots are summed block by
    block, so the iterates equal the serial solver's to rounding, not bit for bit.

    sh: a RowShardedSpMV of the (square) matrix.  x0 / b: FULL vectors (torch, on the rank's device); the solution comes
    back gathered (`x()`)."""

    def __init__(self, sh, x0, b, soft_restart_threshold=0.1):
        if sh.rows != sh.cols or x0.numel() != sh.cols or b.numel() != sh.rows:
            raise ValueError("Dimension mismatch")
        self.sh, self.group = sh, sh.group
        self.r0, self.r1 = sh.r0, sh.r1
        self.full = torch.zeros(sh.cols, dtype=torch.float64, device=x0.device)      # the gathered operand of an SpMV
        self.iteration_count = self.soft_restart_count = self.hard_restart_count = 0
        self.soft_restart_threshold = soft_restart_threshold
        self.b = b[self.r0:self.r1].clone()
        self.x = x0[self.r0:self.r1].clone()
        self.r = self.b - self._matvec(self.x)                       # new(): r = b - A x0 (bicgstab.rs:124)
        self.rhat = self.r.clone()
        self.p = self.r.clone()
        self.err = self._dot(self.r, self.r) ** 0.5
        self.rho = self.err * self.err

    # ---- the two distributed primitives -----------------------------------------------------------------------------
    def _matvec(self, v_block):
        """(A v)[r0:r1] from the rank's block of v: all-gather-v of v, then the local row-block multiply"""
        sh = self.sh
        sh.y[self.r0:self.r1].copy_(v_block)          # the exchange gathers sh.y: borrow it for the operand
        sh.exchange()
        self.full.copy_(sh.y)
        out = torch.empty_like(v_block)
        sh.local_spmv(sh.block, self.full, out)
        return out

    def _dot(self, u, v):
        t = torch.dot(u, v).reshape(1)
        if self.sh.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        # IEEE scalars (numpy float64), not Python floats: an exact breakdown or exact convergence (r == 0, A = I: rho / 0, 0 / 0)
        # gives inf / NaN and the iteration goes on to Err / Ok like the reference (bicgstab.rs:193-223), where a Python float
        # would raise ZeroDivisionError
        return np.float64(t.item())

    # ---- the reference's solver, operand for operand ------------------------------------------------------------------
    def soft_restart(self):
        self.soft_restart_count += 1
        self.rhat = self.r.clone()
        self.rho = self.err * self.err
        self.p = self.r.clone()

    def hard_restart(self):
        self.hard_restart_count += 1
        self.r = self.b - self._matvec(self.x)
        self.err = np.sqrt(self._dot(self.r, self.r))
        self.soft_restart()
        self.soft_restart_count -= 1

    def step(self):
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            return self._step()

    def _step(self):
        self.iteration_count += 1
        v = self._matvec(self.p)
        alpha = self.rho / self._dot(self.rhat, v)
        h = self.x + self.p * alpha
        s = self.r - v * alpha
        t = self._matvec(s)
        omega = self._dot(t, s) / self._dot(t, t)
        self.x = h + s * omega
        self.r = s - t * omega
        self.err = np.sqrt(self._dot(self.r, self.r))
        rho_prev = self.rho
        self.rho = self._dot(self.rhat, self.r)
        if abs(self.rho) / (self.err * self.err) < self.soft_restart_threshold:
            self.soft_restart()
        else:
            beta = (self.rho / rho_prev) * (alpha / omega)
            self.p = self.r + (self.p - v * omega) * beta
        return self.err

    @classmethod
    def solve(cls, sh, x0, b, tol, max_iter, soft_restart_threshold=0.1):
        """BiCGSTAB::solve (bicgstab.rs:148-171); `converged` tells Ok from Err (iteration limit reached, results kept)"""
        solver = cls(sh, x0, b, soft_restart_threshold)
        solver.converged = False
        for _ in range(max_iter):
            solver.step()
            if solver.err < tol:
                solver.hard_restart()             # the true residual, before convergence is claimed
                if solver.err < tol:
                    solver.converged = True
                    break
        return solver

    def x_full(self):
        """the solution, gathered on every rank"""
        sh = self.sh
        sh.y[self.r0:self.r1].copy_(self.x)
        sh.exchange()
        return sh.y.clone()


def hip_local_spgemm(a_block, b):
    """the HIP SpGEMM on torch-resident operands (borrowed, no copy); returns a DeviceCsMat"""
    from . import smmp
    from .device import DeviceCsMat
    da = DeviceCsMat.wrap_torch(a_block[0], a_block[1], a_block[2], a_block[3])
    db = DeviceCsMat.wrap_torch(b[0], b[1], b[2], b[3])
    return smmp.mul_csr_csr(da, db)


class RowShardedSpGEMM:
    """C = A * B with A split into row blocks over the process group, B replicated — the rows of C are
    independent (`smmp::numeric` doc, sprs/src/sparse/smmp.rs:136-140), so there is NO exchange: rank g
    computes and keeps C[r_g:r_{g+1}, :].  The blocks are balanced by the number of PRODUCTS
    sum_{k in A_i} nnz(B_k) (what the SpGEMM kernels' time follows), not by nnz(A).

    a / b: (shape, indptr, indices, data) torch tensors on this rank's device (every rank holds or
    generates the same operands); local_spgemm(a_block, b) -> rank-local result, by default the HIP path
    on the GPU box.  `gather_indptr()` is the one optional collective: the global row pointer of C (the
    prefix sum over ranks of the local nnz, as mul_csr_csr_with_workspace concatenates its chunks,
    smmp.rs:320-331)."""

    def __init__(self, a, b, local_spgemm=None, group=None, virtual=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if virtual is not None:                       # (rank, world) without a process group: one-GPU tests
            self.rank, self.world = virtual
        if local_spgemm is None:
            local_spgemm = hip_local_spgemm
        (self.rows, self.inner), a_ip, a_ix, a_dt = a
        (b_rows, self.cols), b_ip, _, _ = b
        if self.inner != b_rows:
            raise ValueError("Dimension mismatch")                       # smmp.rs:207
        # products per row of A, and their prefix: the cost the blocks are balanced on
        b_len = (b_ip[1:] - b_ip[:-1]).to(torch.float64)
        per_entry = b_len[a_ix.long()]
        cost = torch.zeros(self.rows + 1, dtype=torch.float64, device=a_ip.device)
        csum = torch.cumsum(per_entry, 0)
        ends = (a_ip[1:] - a_ip[0]).long()
        cost[1:] = torch.where(ends > 0, csum[(ends - 1).clamp(min=0)], torch.zeros((), dtype=torch.float64,
                                                                                     device=a_ip.device))
        cost[1:] = torch.cummax(cost[1:], 0).values                      # empty rows inherit the running total
        self.cuts = gen.balanced_row_blocks(cost.to(torch.float64), self.world)
        r0, r1 = self.cuts[self.rank], self.cuts[self.rank + 1]
        lo, hi = int(a_ip[r0] - a_ip[0]), int(a_ip[r1] - a_ip[0])
        self.r0, self.r1 = r0, r1
        self.block_products = float(cost[r1] - cost[r0])
        self.a_block = ((r1 - r0, self.inner), (a_ip[r0:r1 + 1] - a_ip[r0]).contiguous(), a_ix[lo:hi].clone(),
                        a_dt[lo:hi].clone())
        self.b = b
        self.local_spgemm = local_spgemm
        self.c_block = None

    def multiply(self):
        """rank-local C[r0:r1, :]"""
        self.c_block = self.local_spgemm(self.a_block, self.b)
        return self.c_block

    def gather_indptr(self, local_indptr):
        """global indptr of C from the ranks' zero-based local ones (a device/CPU int64 tensor of r1-r0+1)."""
        nnz_local = torch.tensor([int(local_indptr[-1])], dtype=torch.int64, device=local_indptr.device)
        if self.world == 1:
            return local_indptr.clone()
        all_nnz = [torch.zeros_like(nnz_local) for _ in range(self.world)]
        dist.all_gather(all_nnz, nnz_local, group=self.group)
        base = sum(int(v) for v in all_nnz[:self.rank])
        pieces = [torch.zeros(self.cuts[g + 1] - self.cuts[g] + 1, dtype=torch.int64, device=local_indptr.device)
                  for g in range(self.world)]
        # variable lengths: pad to the longest block for the collective
        pad = max(p.numel() for p in pieces)
        mine = torch.zeros(pad, dtype=torch.int64, device=local_indptr.device)
        mine[:local_indptr.numel()] = local_indptr.to(torch.int64) + base
        got = [torch.zeros(pad, dtype=torch.int64, device=local_indptr.device) for _ in range(self.world)]
        dist.all_gather(got, mine, group=self.group)
        out = [got[g][:pieces[g].numel()] for g in range(self.world)]
        return torch.cat([out[0]] + [o[1:] for o in out[1:]])
