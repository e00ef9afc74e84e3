#!/usr/bin/env python3
"""Developer printout: per-workgroup timeline of the banded SpMV's hot kernel (needs the DEVTOOLS library:
SPRS_HIP_LIBRARY=sprs_amd/libsprs_hip_dev.so SPRS_HIP_HOTPROF=1).  usage: hot_prof.py [workload] [opt=val ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sprs_amd
from sprs_amd import gen, prod
from sprs_amd.device import DeviceCsMat, DeviceVec

wl = sys.argv[1] if len(sys.argv) > 1 else "rmat1m"
n, k = (1_000_000, 16) if wl == "rmat1m" else (10_000_000, 32)
dev = torch.device("cuda", 0)
ip, ix, dt = gen.rmat_csr(n, k, device=dev)
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    sprs_amd.set_option(key, int(val))
sprs_amd.set_option("spmv_band_debug", 16)
a = DeviceCsMat.wrap_torch((n, n), ip, ix, dt).prepare()
x = DeviceVec.borrow(gen.dense_vector(n, seed=3, device=dev))
y = DeviceVec.borrow(torch.empty(n, dtype=torch.float64, device=dev))
print("==", wl, sys.argv[2:], file=sys.stderr)
for _ in range(4):
    prod.csmat_mul_vec(a, x, out=y)
torch.cuda.synchronize()
