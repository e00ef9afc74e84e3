#!/bin/bash
# SpMM with the re-laid-out rhs copy as the default (auto): gate, bench over k, a column-major rhs, then the driver's bench line
mkdir -p gpurun_out/r11zb
timeout 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
timeout 300 python scripts/spmm_bench.py 10000000 32 8 12 16 32 64 2>&1 | grep -v amdgpu | cut -c1-140 | tee gpurun_out/r11zb/spmm_bench.jsonl
timeout 300 python scripts/spmm_colmajor_rhs.py 2>&1 | grep -v amdgpu | tee gpurun_out/r11zb/spmm_colmajor_rhs.jsonl
bash scripts/gpu_session.sh r11zb bench | cut -c1-200
