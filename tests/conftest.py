import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """The reference's golden vectors (tests/golden/make_fixtures.py)."""
    with open(os.path.join(ROOT, "tests", "golden", "sprs_fixtures.json")) as f:
        return json.load(f)


IDX_COMBOS = [(np.uint64, np.uint64), (np.uint32, np.uint32), (np.uint32, np.uint64)]


def as_csr(fx, idx=np.uint64, ptr=np.uint64):
    """fixture dict -> (shape, indptr, indices, data) numpy arrays"""
    return (tuple(fx["shape"]), np.array(fx["indptr"], dtype=ptr),
            np.array(fx["indices"], dtype=idx), np.array(fx.get("data", []), dtype=np.float64))


@pytest.fixture(scope="module", autouse=True)
def _poison_device_memory(request):
    """Before every GPU test module: fill a few GB of device memory with 0xFF bytes (NaN as f64,
    2^64-1 as an index) and release it, so that blocks handed out afterwards are NOT zero.  Fresh
    hipMalloc memory usually reads as zero, which hides kernels that forget to initialise a buffer
    (found that way: SpMV partials of an x-slice without entries, spmv.hip get_scratch)."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("SPRS_HIP_LIBRARY"):
        yield       # (a library given explicitly = the kernel emulator of tests/emu, which poisons every block itself)
        return
    import ctypes as C
    import sprs_amd
    from sprs_amd import _ffi
    if sprs_amd.device_count() >= 1:
        blocks = []
        for nbytes, count in ((64 << 20, 48), (1 << 20, 256), (4096, 2048)):
            for _ in range(count):
                p = C.c_void_p()
                if _ffi.lib.sprs_hip_malloc(C.byref(p), nbytes) != 0:
                    break
                _ffi.lib.sprs_hip_memset(p, 0xFF, nbytes, None)
                blocks.append(p)
        _ffi.lib.sprs_hip_synchronize(None)
        for p in blocks:
            _ffi.lib.sprs_hip_free(p)
    yield
