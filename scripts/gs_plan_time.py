#!/usr/bin/env python3
"""One-time costs of the Gauss-Seidel path on the heat system of a G x G grid: the level order (device), the SpMV plan.
usage: gs_plan_time.py [G]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sprs_amd import gen
from sprs_amd.device import DeviceCsMat, DeviceVec
from sprs_amd.linalg import gauss_seidel
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ip, ix, dt = gen.grid_laplacian(g, g, device=dev)
a = DeviceCsMat.wrap_torch((g * g, g * g), ip, ix, dt)
x = DeviceVec.zeros(g * g); rhs = DeviceVec.zeros(g * g)
y = a * x                      # SpMV plan
torch.cuda.synchronize()
t = time.perf_counter(); gauss_seidel(a, x, rhs, 0, 1e-8); torch.cuda.synchronize(); first = time.perf_counter() - t
t = time.perf_counter(); gauss_seidel(a, x, rhs, 0, 1e-8); torch.cuda.synchronize(); second = time.perf_counter() - t
print(json.dumps({"grid": g, "rows": g * g, "level_order_s": round(first - second, 4), "call_without_plan_s": round(second, 4)}))
