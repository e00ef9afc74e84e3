#!/bin/bash
# round 4, call t: time of the Gauss-Seidel level order on the device
mkdir -p gpurun_out/r10t
timeout 300 python scripts/gs_plan_time.py 4096 2>&1 | grep -v amdgpu | tee gpurun_out/r10t/gs_plan_time.jsonl
timeout 300 python scripts/gs_plan_time.py 1024 2>&1 | grep -v amdgpu | tee -a gpurun_out/r10t/gs_plan_time.jsonl
