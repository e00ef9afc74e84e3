"""CPU model of the row placement of the SpMM's re-laid-out rhs copy (spmm.hip: relaid_row): the kernel that writes the copy
and the kernel that gathers from it use the same function, so all that has to hold is that it is a BIJECTION of every block of 4096 rows
onto itself (no two rows of the rhs may land on one row of the copy, and the copy has no more rows than the padded rhs) — and, for the
function to be worth its cost, that the hub columns of a power-law matrix (0, 2^k, 2^j + 2^k) do not stay power-of-two strides apart."""
import re
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def constants():
    src = open(os.path.join(ROOT, "sprs_amd", "csrc", "spmm.hip")).read()
    bits = int(re.search(r"RL_BITS = (\d+)", src).group(1))
    m = re.search(r"\(\(c \* (0x[0-9A-Fa-f]+)u \+ hi \* (0x[0-9A-Fa-f]+)u\) & RL_MASK\)", src)
    return bits, int(m.group(1), 16), int(m.group(2), 16)


def relaid_row(c, bits, mul, rot):
    c = np.asarray(c, dtype=np.uint64)
    hi = c >> np.uint64(bits)
    mask = np.uint64((1 << bits) - 1)
    return (hi << np.uint64(bits)) | (((c * np.uint64(mul) + hi * np.uint64(rot)) & np.uint64(0xFFFFFFFF)) & mask)


def test_every_block_is_permuted_onto_itself():
    bits, mul, rot = constants()
    assert mul % 2 == 1                                           # an odd multiplier is a bijection modulo a power of two
    n = 5 << bits
    img = relaid_row(np.arange(n), bits, mul, rot)
    assert np.array_equal(np.sort(img), np.arange(n, dtype=np.uint64))
    assert np.array_equal(img >> np.uint64(bits), np.arange(n, dtype=np.uint64) >> np.uint64(bits))
    top = np.arange((1 << 32) - (1 << bits), 1 << 32, dtype=np.uint64)          # the last block 32-bit column ids can name
    assert np.array_equal(np.sort(relaid_row(top, bits, mul, rot)), top)


def test_hub_columns_are_spread():
    bits, mul, rot = constants()
    hubs = np.array(sorted({0} | {1 << k for k in range(bits)} | {(1 << j) + (1 << k) for j in range(bits) for k in range(j)}), dtype=np.uint64)
    img = relaid_row(hubs, bits, mul, rot)
    # 128-byte rows: which of 16 / 64 interleaved channels a row falls on, if row-index bits 4 and up pick the channel (address bits 11
    # and up: an odd multiplier carries low bits upward, it does not scramble the lowest ones — the measured effect, 7.9 -> 5.9 ms on the
    # hardware, says those are the bits that matter; profiles/r11za)
    for ch in (16, 64):
        before = np.bincount(((hubs >> np.uint64(4)) % ch).astype(np.int64), minlength=ch)
        after = np.bincount(((img >> np.uint64(4)) % np.uint64(ch)).astype(np.int64), minlength=ch)
        assert after.max() * 2 <= before.max(), (ch, before.max(), after.max())
