#!/bin/bash
mkdir -p gpurun_out/r11i
timeout 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
timeout 600 python scripts/spmm_bench.py 10000000 32 8 16 32 2>&1 | grep -v amdgpu | tee gpurun_out/r11i/spmm_bench.jsonl
