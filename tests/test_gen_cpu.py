"""CPU checks of the synthetic generators (sprs_amd/gen.py) against the oracle's
reference-faithful constructions."""
import numpy as np
import torch

from oracle import oracle
from sprs_amd import gen


def test_grid_laplacian_matches_heat_example():
    # sprs/examples/heat.rs:45-80 via the oracle's literal restatement
    for r in (3, 4, 17):
        ip, ix, dt = gen.grid_laplacian(r, r)
        shape, oip, oix, odt = oracle.grid_laplacian(r, r)
        assert np.array_equal(ip.numpy(), oip.astype(np.int64))
        assert np.array_equal(ix.numpy(), oix.astype(np.int64))
        assert np.array_equal(dt.numpy(), odt)


def test_rmat_is_valid_sorted_deduped_and_deterministic():
    n = 30000
    ip, ix, dt = gen.rmat_csr(n, 8, seed=1)
    oracle.check_structure(n, n, ip.numpy().astype(np.uint64), ix.numpy().astype(np.uint64))  # strictly increasing rows
    assert 0.5 <= float(dt.min()) and float(dt.max()) < 1.5
    lens = np.diff(ip.numpy())
    assert 6.5 < lens.mean() < 9.5 and (lens == 0).mean() > 0.2 and lens.max() > 50 * lens.mean()   # power law
    ip2, ix2, dt2 = gen.rmat_csr(n, 8, seed=1, chunk=1 << 14)
    assert torch.equal(ip, ip2) and torch.equal(ix, ix2) and torch.equal(dt, dt2)
    ip3, ix3, _ = gen.rmat_csr(n, 8, seed=2)
    assert not torch.equal(ix, ix3[: ix.numel()]) if ix3.numel() >= ix.numel() else True


def test_index_width_variants():
    ip, ix, dt = gen.rmat_csr(5000, 4, idx_dtype=torch.int32, ptr_dtype=torch.int32)
    ip8, ix8, dt8 = gen.rmat_csr(5000, 4)
    assert ix.dtype == torch.int32 and ip.dtype == torch.int32
    assert torch.equal(ix.to(torch.int64), ix8) and torch.equal(ip.to(torch.int64), ip8) and torch.equal(dt, dt8)


def test_balanced_row_blocks():
    ip, ix, dt = gen.rmat_csr(40000, 8)
    for parts in (1, 2, 4, 8):
        cuts = gen.balanced_row_blocks(ip, parts)
        assert cuts[0] == 0 and cuts[-1] == 40000 and len(cuts) == parts + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        nnz = [int(ip[b] - ip[a]) for a, b in zip(cuts, cuts[1:])]
        assert sum(nnz) == ix.numel()
        longest = int(np.diff(ip.numpy()).max())
        assert max(nnz) <= ix.numel() / parts + longest


def test_uniform_csr_is_the_reference_generators_distribution():
    # sprs-rand/src/lib.rs:24-88 (rand_csr) and its own test random_csr (lib.rs:105-113): density within 0.25 .. 0.35
    ip, ix, dt = gen.uniform_csr((100, 70), 0.3)
    assert ix.numel() == int(np.ceil(0.3 * 100 * 70))                    # exactly exp_nnz entries (lib.rs:37-38, 64)
    oracle.check_structure(70, 100, ip.numpy().astype(np.uint64), ix.numpy().astype(np.uint64))   # rows strictly increasing: no repeated column
    assert 0.25 < ix.numel() / 7000.0 < 0.35
    ip0, ix0, dt0 = gen.uniform_csr((0, 0), 0.3)                         # lib.rs:98-103 empty_random_mat
    assert ix0.numel() == 0 and ip0.numel() == 1
    # the bench shape rule: nnz_over_rows = 4 -> density 4 / cols (sprs-benches/src/main.rs:178-186)
    n = 20000
    ip, ix, dt = gen.uniform_csr((n, n), 4.0 / n, seed=5)
    lens = np.diff(ip.numpy())
    assert ix.numel() == 4 * n and abs(lens.mean() - 4.0) < 1e-9 and lens.max() < 20          # multinomial rows, no power law
    assert abs(float(dt.mean())) < 0.02 and abs(float(dt.std()) - 1.0) < 0.02                  # N(0, 1) values
    cols = np.bincount(ix.numpy(), minlength=n)
    assert cols.max() < 25                                                                     # uniform columns
    ip2, ix2, dt2 = gen.uniform_csr((n, n), 4.0 / n, seed=5)
    assert torch.equal(ip, ip2) and torch.equal(ix, ix2) and torch.equal(dt, dt2)
    # a dense-ish small case forces the redraw rounds
    ip, ix, dt = gen.uniform_csr((50, 24), 0.3, seed=9)
    oracle.check_structure(24, 50, ip.numpy().astype(np.uint64), ix.numpy().astype(np.uint64))
    assert ix.numel() == 360
