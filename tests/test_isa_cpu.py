"""Guards of two compiler-level facts this path depends on (hipcc cross-compiles without a GPU):

* the entry loads of a chunk / batch of the SpGEMM window kernels go out back to back.  A load under a per-lane condition is
  compiled as a branch whose arm ends in `s_waitcnt vmcnt(0)` for that one load; rounds 1 - 3 shipped kernels whose "independent"
  loads were four / eight memory round trips one after the other and nothing but the ISA shows it;
* the row loop of the wave-per-row kernels draws its rows in the loop HEADER (`for (q = draw(); q < n; q = draw())`): written as
  `for (;;) { q = draw(); if (q >= n) break; ... }` one build variant was compiled into a loop that never ended on config 5.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def spgemm_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "spgemm.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                           os.path.join(ROOT, "sprs_amd", "csrc", "spgemm.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN8sprs_hip\S+):\s*; @\S+\n(.*?)^\s*s_endpgm", text, flags=re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    return kernels


def longest_load_burst(body):
    """most global loads issued without a wait for memory in between"""
    best = cur = 0
    for line in body.splitlines():
        ins = line.strip().split(" ")[0]
        if ins.startswith("global_load"):
            cur += 1
            best = max(best, cur)
        elif ins == "s_waitcnt" and "vmcnt" in line:
            cur = 0
    return best


def pick(kernels, *parts):
    names = [k for k in kernels if all(p in k for p in parts)]
    assert names, parts
    return names


def test_chunk_loads_of_the_wave_kernels_are_issued_together(spgemm_isa):
    # numeric, usize / usize, windows of 2^15, chunks of 8 wave instructions: 8 records = 16 loads (column dword + value dwordx2)
    for name in pick(spgemm_isa, "mid_rows_kernelImmLb1ELi15ELi8E"):
        assert longest_load_burst(spgemm_isa[name]) >= 16, name
    # counting twin: 8 column loads
    for name in pick(spgemm_isa, "mid_rows_kernelImmLb0ELi14ELi8E"):
        assert longest_load_burst(spgemm_isa[name]) >= 8, name


def test_batch_loads_of_the_workgroup_kernel_are_issued_together(spgemm_isa):
    # numeric: 4 records per batch = 8 loads; counting: 4 column loads
    for name in pick(spgemm_isa, "large_rows_kernelILi17EmmLb1ELi6E"):
        assert longest_load_burst(spgemm_isa[name]) >= 8, name
    for name in pick(spgemm_isa, "large_rows_kernelILi18EmmLb0ELi1E"):
        assert longest_load_burst(spgemm_isa[name]) >= 4, name


def test_row_draw_sits_in_the_loop_header():
    src = open(os.path.join(ROOT, "sprs_amd", "csrc", "spgemm.hip")).read()
    assert "for (uint64_t q = draw_row(); q < n_mid; q = draw_row())" in src
    body = src[src.index("void mid_rows_kernel("):src.index("// ranks of the batch's columns")]
    body = re.sub(r"//[^\n]*", "", body)                     # (the comment that tells the story quotes the bad form)
    assert "for (;;)" not in body, "a wave-uniform draw inside `for (;;)` + `break` was mis-compiled once (DESIGN 4.2)"


@pytest.fixture(scope="module")
def spmm_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "spmm.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                           os.path.join(ROOT, "sprs_amd", "csrc", "spmm.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN8sprs_hip\S+):\s*; @\S+\n(.*?)^\s*s_endpgm", text, flags=re.S | re.M):
        meta = text[text.index(".amdhsa_kernel " + m.group(1)):]
        kernels[m.group(1)] = (m.group(2), int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1)),
                               int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1)))
    return kernels


def test_spmm_stream_kernel_keeps_its_gathers_in_flight(spmm_isa):
    """the entry-stream SpMM kernel: 16 rhs rows in flight per lane behind ONE wait sequence (a gather under a per-lane
    condition would be a round trip each), no scratch, and few enough registers for 6 waves per SIMD (7 workgroups of 20 KB LDS per CU)"""
    names = [k for k in spmm_isa if "spmm_stream_kernel" in k]
    assert len(names) == 4 * 4 * 2 * 2                          # index x pointer widths, KP = 8 .. 64, accumulate, re-laid-out rhs
    for name in names:
        body, vgpr, scratch = spmm_isa[name]
        assert scratch == 0, name
        assert vgpr <= 84, (name, vgpr)
        assert longest_load_burst(body) >= 16, name


@pytest.fixture(scope="module")
def band_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "spmv_band.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                           os.path.join(ROOT, "sprs_amd", "csrc", "spmv_band.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN8sprs_hip\S+):\s*; @\S+\n(.*?)^\s*s_endpgm", text, flags=re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    return kernels, text


def test_hot_kernel_does_not_drain_its_stores_before_the_next_request(band_isa):
    """gfx950 counts loads and stores on ONE in-order vmcnt.  Until round 5 the compiler sank the use of the tile's last load
    (tile_row[w]) behind the flush of the previous tile's sums: `global_store ... ; s_waitcnt vmcnt(0) ; v_mov ; global_load ...` —
    every wave waited for the acknowledgement of its stores with nothing in flight, once per tile (hot kernel alone 736 us; 658 us
    with the wait in front of the flush, profiles/r13c).  Guard: in the loop of band_hot_kernel no full drain sits between the
    non-temporal store of the flush and the next non-temporal tile loads; and the kernel keeps its 96 VGPRs (two 64-register
    gather waves fit beside each of its waves) without scratch."""
    kernels, text = band_isa
    for name in pick(kernels, "band_hot_kernelILi14E"):
        lines = [ln.strip() for ln in kernels[name].splitlines()]
        stores = [i for i, ln in enumerate(lines) if ln.startswith("global_store_dwordx2") and ln.endswith(" nt")]
        assert stores, name
        first = stores[0]                                    # the deferred flush of the previous tile: the first nt store of the loop
        nxt = next(i for i in range(first, len(lines)) if lines[i].startswith("global_load_dwordx4") and lines[i].endswith(" nt"))
        between = lines[first:nxt]
        assert not any(ln.startswith("s_waitcnt") and "vmcnt(0)" in ln for ln in between), \
            "%s: the flush's stores are drained before the next tile is requested:\n%s" % (name, "\n".join(between))
        tail = text[text.index(name + ":"):]
        m = re.search(r"; NumVgprs: (\d+).*?; ScratchSize: (\d+)", tail, flags=re.S)
        assert m and int(m.group(1)) <= 96 and int(m.group(2)) == 0, (name, m and m.groups())
