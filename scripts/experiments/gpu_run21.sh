#!/bin/bash
# SpGEMM v2 (entry-parallel expansion + LDS sort): parity tests, then config 5 with and without phase profile
mkdir -p gpurun_out/r01z
{
timeout 900 python -m pytest tests/test_spgemm_gpu.py tests/test_convert_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "--- config5 plain"
timeout 600 python tests/spgemm_bench.py 1000000 8 8 300
echo "--- config5 prof"
SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 300 2>&1 | tail -12
} 2>&1 | tee gpurun_out/r01z/log.txt
