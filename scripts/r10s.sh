#!/bin/bash
# round 4, call s: Gauss-Seidel level order computed on the device (no structure download, no host pass)
bash scripts/gpu_session.sh r10s "gate:gauss_seidel" "py:scripts/gauss_seidel_bench.py 4096 3"
