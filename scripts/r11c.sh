#!/bin/bash
# SpMM: fills or walk?  the chunk kernel on an L2-resident rhs, on uniform columns and on R-MAT
mkdir -p gpurun_out/r11c
timeout 600 python scripts/spmm_bound_probe.py 8 16 2>&1 | grep -v amdgpu | tee gpurun_out/r11c/spmm_bound_probe.jsonl
