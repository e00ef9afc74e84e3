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
