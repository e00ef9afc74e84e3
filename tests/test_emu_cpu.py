"""Kernel LOGIC on the CPU: the SpGEMM kernels compiled for the host and run by the fiber emulator of tests/emu, with the
waves of a workgroup scheduled in three different orders.  The large-row value kernel hands a token from wave to wave
(spgemm.hip, "ORDER OF THE ADDITIONS"); its results must not depend on which wave runs first.  This is test
infrastructure: the emulator is never loaded by the product and says nothing about speed; parity on the real gfx950
build is the job of the -m gpu tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emu_lib():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ of the ROCm toolchain here")
    r = subprocess.run(["make", "-C", EMU, "-j8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(EMU, "libsprs_hip_emu.so")


def test_lane_exchanges_of_the_emulator(emu_lib):
    """The shipped kernels' lane exchanges (sprs_amd/csrc/lanes.hpp and the DPP scan ladders: quad_perm, row_shr / row_shl /
    row_ror, wave_shr / wave_shl, row_bcast 15 / 31 with row and bank masks, v_permlane16_swap / v_permlane32_swap) run in the
    emulator AS WRITTEN — there is no __shfl alternative in the product sources — so the emulator's model of those controls is
    itself checked here, by the probe that checks the hardware (scripts/probes/lane_ops.hip: every exchange against index
    arithmetic, the group scans against a loop, the group sorts against std::sort)."""
    r = subprocess.run(["make", "-C", EMU, "lane_ops_emu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([os.path.join(EMU, "lane_ops_emu"), "256"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and '"ok": true' in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("order", ["default", "reverse", "rotate"])
def test_spgemm_kernels_under_wave_orders(emu_lib, order):
    env = dict(os.environ, SPRS_HIP_LIBRARY=emu_lib, HIPEMU_WAVE_ORDER=order)
    sel = "golden_mul_csr_csr or zero_rows or structural_zeros or rectangular or multi_window or order_of_additions or class_boundaries or micro_rows or hub_rows"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_spgemm_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-k", sel, "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_spmv_band_kernels_in_the_emulator(emu_lib):
    """the banded SpMV plan (hot slices from an LDS tile, cold pieces, split permutation, carries, reduction) and its plan
    build, on the small matrices of tests/test_spmv_band_gpu.py"""
    env = dict(os.environ, SPRS_HIP_LIBRARY=emu_lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_spmv_band_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("order", ["default", "reverse", "rotate"])
def test_gauss_seidel_sweep_in_the_emulator(emu_lib, order):
    """the sync-free sweep kernel (rows in level order, values handed from wave to wave through the next iterate) on the
    small systems of tests/test_gauss_seidel_gpu.py, with the waves of a workgroup scheduled in three orders"""
    env = dict(os.environ, SPRS_HIP_LIBRARY=emu_lib, HIPEMU_WAVE_ORDER=order)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gauss_seidel_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_spmm_stream_kernel_in_the_emulator(emu_lib):
    """the entry-stream SpMM kernel (tiles of 256 entries, rows by marks + max-scan, runs per lane group, fix-up of the rows that
    cross tiles) at its seams, and the reference's golden dense products, on the CPU"""
    env = dict(os.environ, SPRS_HIP_LIBRARY=emu_lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_spmm_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-k", "stream_tiles or golden or ragged", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_differential_fuzzer_in_the_emulator(emu_lib):
    """scripts/fuzz_parity.py (nine kinds of seeded random cases against the oracle; on the GPU: profiles/r08g, r09c) for
    twenty seconds through the emulated kernels: the script itself stays runnable and the kernels' logic agrees on whatever
    shapes the seeds of this run produce"""
    env = dict(os.environ, SPRS_HIP_LIBRARY=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "20", "777"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert '"failures": 0' in r.stdout
