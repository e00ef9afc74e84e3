"""Deterministic synthetic matrices for tests and bench.py (the role sprs-rand
plays for the reference's benches, sprs-rand/src/lib.rs:24-88 — which has no
R-MAT generator, SURVEY F8, so the generator is defined here).

Harness code, not the product: it uses torch only as an array library so that
the SAME generator runs on the CPU (tests, small sizes) and on the GPU (bench
sizes: 3.2e8 edges take ~1 s there instead of minutes of numpy).  Randomness is
a counter-based SplitMix64 hash, so results do not depend on the device, on
chunking or on the torch version.

Definitions:
  R-MAT      Graph500 quadrant probabilities (a,b,c,d) = (.57,.19,.19,.05),
             scale = ceil(log2 n); edges with row >= n or col >= n rejected,
             duplicates DROPPED (not summed), rows sorted; values U[0.5,1.5).
             Each level consumes 16 bits of a SplitMix64 word
             (thresholds round(p * 65536)).
  Laplacian  grid_laplacian of sprs/examples/heat.rs:45-80.
  uniform    the matrices of the reference's own benches (sprs-rand `rand_csr`,
             sprs-rand/src/lib.rs:24-88; shapes of sprs-benches/src/main.rs:102-164):
             ceil(density * rows * cols) entries, the row of each drawn uniformly,
             the columns of a row drawn uniformly WITHOUT repetition and sorted,
             values standard normal.  Same distribution as the reference's generator,
             not the same stream (its rng is Pcg64Mcg seeded from the OS).
"""
import math

import torch

_MASK64 = (1 << 64) - 1


def _s64(v):
    """python int (mod 2^64) -> the int64 with the same bits"""
    v &= _MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


_GOLDEN = _s64(0x9E3779B97F4A7C15)
_M1 = _s64(0xBF58476D1CE4E5B9)
_M2 = _s64(0x94D049BB133111EB)


def _lsr(z, k):
    """logical shift right of int64 tensors"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64(counter, seed):
    """counter: int64 tensor; returns int64 tensor of hashed bits (two's complement)."""
    z = counter * _GOLDEN + _s64(seed * 0xD1342543DE82EF95 + 0x2545F4914F6CDD1D)
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def uniform_05_15(counter, seed):
    """U[0.5, 1.5) doubles from 53 hashed bits."""
    return _lsr(splitmix64(counter, seed), 11).to(torch.float64) * (1.0 / (1 << 53)) + 0.5


def dense_vector(n, seed=3, device="cpu"):
    """x ~ U[0.5,1.5): positive, so relative error checks are meaningful."""
    return uniform_05_15(torch.arange(n, dtype=torch.int64, device=device), seed)


_TA = round(0.57 * 65536)            # u < TA            -> (0,0)
_TB = round((0.57 + 0.19) * 65536)   # TA <= u < TB      -> (0,1)
_TC = round((0.57 + 0.38) * 65536)   # TB <= u < TC      -> (1,0)   else (1,1)


def rmat_csr(n, nnz_per_row, seed=1, value_seed=2, oversample=None, device="cpu",
             idx_dtype=torch.int64, ptr_dtype=torch.int64, chunk=1 << 26):
    """R-MAT CSR matrix, n x n, ~nnz_per_row stored entries per row.

    Draws ceil(n * nnz_per_row * oversample) raw edges; rejection + dedupe lose
    ~4 % at n = 1e6 and ~22 % at n = 1e7, hence the default
    oversample of 1.05 / 1.30.  Returns (indptr, indices, data) torch tensors on
    `device` (int64 bit patterns == sprs' usize; int32 == u32)."""
    scale = max(1, math.ceil(math.log2(n)))
    if oversample is None:
        oversample = 1.30 if n >= 5_000_000 else 1.05
    draws = int(math.ceil(n * nnz_per_row * oversample))
    words = (scale + 3) // 4                     # 4 levels (16 bits each) per hash word
    keys = []
    for lo in range(0, draws, chunk):
        hi = min(draws, lo + chunk)
        e = torch.arange(lo, hi, dtype=torch.int64, device=device)
        row = torch.zeros_like(e)
        col = torch.zeros_like(e)
        level = 0
        for w in range(words):
            h = splitmix64(e * words + w, seed)
            for q in range(4):
                if level == scale:
                    break
                u = _lsr(h, 16 * q) & 0xFFFF
                rbit = (u >= _TB).to(torch.int64)
                cbit = (((u >= _TA) & (u < _TB)) | (u >= _TC)).to(torch.int64)
                row = (row << 1) | rbit
                col = (col << 1) | cbit
                level += 1
        ok = (row < n) & (col < n)
        keys.append(((row << 32) | col)[ok])
        del e, row, col, h, u, rbit, cbit, ok
    key = torch.cat(keys) if len(keys) > 1 else keys[0]
    del keys
    key = torch.unique(key, sorted=True)          # sort by (row, col) + drop duplicates
    rows = key >> 32
    indices = (key & 0xFFFFFFFF).to(idx_dtype)
    counts = torch.bincount(rows, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=indptr[1:])
    del rows, counts, key
    nnz = indices.numel()
    data = uniform_05_15(torch.arange(nnz, dtype=torch.int64, device=device), value_seed)
    return indptr.to(ptr_dtype), indices, data


def standard_normal(counter, seed):
    """N(0,1) doubles: Box-Muller on two hashed uniforms (deterministic, device-independent up to libm rounding)."""
    u1 = (_lsr(splitmix64(counter * 2, seed), 11).to(torch.float64) + 1.0) * (1.0 / (1 << 53))     # (0, 1]
    u2 = _lsr(splitmix64(counter * 2 + 1, seed), 11).to(torch.float64) * (1.0 / (1 << 53))
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)


def uniform_csr(shape, density, seed=11, value_seed=12, device="cpu", idx_dtype=torch.int64, ptr_dtype=torch.int64):
    """The role of sprs-rand's rand_csr (sprs-rand/src/lib.rs:24-88): exp_nnz = ceil(density * rows * cols) stored
    entries; every entry draws its ROW uniformly (rows end up multinomially filled, lib.rs:44-63), then every row draws
    that many DISTINCT columns uniformly (the reference redraws a column that is already present, lib.rs:68-77 — here all
    clashes of a round are redrawn together until none is left) and sorts them; values ~ N(0,1) (rand_csr_std).
    Returns (indptr, indices, data) torch tensors on `device`."""
    rows, cols = int(shape[0]), int(shape[1])
    assert 0.0 <= density <= 1.0
    nnz = int(math.ceil(density * rows * cols))
    indptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    if nnz == 0 or rows == 0 or cols == 0:
        return indptr.to(ptr_dtype), torch.zeros(0, dtype=idx_dtype, device=device), torch.zeros(0, dtype=torch.float64, device=device)
    e = torch.arange(nnz, dtype=torch.int64, device=device)
    row = _lsr(splitmix64(e, seed), 1) % rows
    row, _ = torch.sort(row)
    counts = torch.bincount(row, minlength=rows)
    if int(counts.max()) > cols:
        raise ValueError("a row drew more entries than there are columns (density too high for this generator)")
    torch.cumsum(counts, 0, out=indptr[1:])
    col = _lsr(splitmix64(e, seed + 1), 1) % cols
    rnd = 0
    while True:
        key, order = torch.sort((row << 32) | col)          # rows are sorted already: this sorts the columns inside each row
        col = key & 0xFFFFFFFF
        dup = torch.zeros(nnz, dtype=torch.bool, device=device)
        dup[1:] = key[1:] == key[:-1]
        ndup = int(dup.sum())
        if ndup == 0:
            break
        rnd += 1
        idx = torch.nonzero(dup).flatten()
        col[idx] = _lsr(splitmix64(idx + nnz * rnd, seed + 1), 1) % cols
    data = standard_normal(e, value_seed)
    return indptr.to(ptr_dtype), col.to(idx_dtype), data


def grid_laplacian(rows, cols, device="cpu", idx_dtype=torch.int64, ptr_dtype=torch.int64):
    """grid_laplacian (sprs/examples/heat.rs:45-80), vectorised: border
    vertices carry a single diagonal 1.0, interior ones (1,1,-4,1,1) at columns
    (i-1,j),(i,j-1),(i,j),(i,j+1),(i+1,j), flattened as i*rows+j like heat.rs:60."""
    nv = rows * cols
    v = torch.arange(nv, dtype=torch.int64, device=device)
    i, j = v // cols, v % cols
    border = (i == 0) | (i == rows - 1) | (j == 0) | (j == cols - 1)
    counts = torch.where(border, 1, 5)
    indptr = torch.zeros(nv + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = torch.empty(nnz, dtype=torch.int64, device=device)
    data = torch.empty(nnz, dtype=torch.float64, device=device)
    start = indptr[:-1]
    flat = i * rows + j
    bs = start[border]
    indices[bs] = flat[border]
    data[bs] = 1.0
    inner = ~border
    s, ii, jj = start[inner], i[inner], j[inner]
    for k, (di, dj, val) in enumerate(((-1, 0, 1.0), (0, -1, 1.0), (0, 0, -4.0), (0, 1, 1.0), (1, 0, 1.0))):
        indices[s + k] = (ii + di) * rows + (jj + dj)
        data[s + k] = val
    return indptr.to(ptr_dtype), indices.to(idx_dtype), data


def balanced_row_blocks(indptr, parts, row_weight=0.0):
    """Contiguous row blocks of ~equal COST, cost(row) = nnz(row) + row_weight:
    boundaries r_0=0 <= r_1 <= ... <= r_parts = rows, r_g = first row whose cumulative cost
    reaches g/parts of the total.  row_weight = 0 balances stored entries only; the per-row
    constant (indptr read, y write, a segment to reduce) is worth ~5 entries on MI355X
    (measured with scripts/virtual_ranks.py: blocks of equal nnz ran 0.245 .. 0.35 ms).
    Each block is exactly A.slice_outer(r_g..r_{g+1}) of the reference (slicing.rs:65-89)."""
    rows = indptr.numel() - 1
    cost = (indptr - indptr[0]).to(torch.float64)
    if row_weight:
        cost = cost + row_weight * torch.arange(rows + 1, dtype=torch.float64, device=indptr.device)
    total = float(cost[-1])
    targets = torch.tensor([total * g / parts for g in range(parts + 1)], dtype=torch.float64,
                           device=indptr.device)
    cuts = torch.searchsorted(cost[:-1].contiguous(), targets, right=False)
    cuts[0] = 0
    cuts[-1] = rows
    return [int(c) for c in cuts]
