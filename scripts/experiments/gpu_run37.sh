#!/bin/bash
mkdir -p gpurun_out/r01zb
{
timeout 900 python -m pytest tests/test_bicgstab_gpu.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python scripts/bicgstab_bench.py 4096 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/bicgstab_bench.py 1024 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r01zb/log.txt
