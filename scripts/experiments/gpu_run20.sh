#!/bin/bash
TAG=${1:-r01w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spmm + convert"
timeout 900 python -m pytest tests/test_spmm_gpu.py tests/test_convert_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "== spmm bench rmat10m"
timeout 900 python scripts/spmm_bench.py 10000000 32 8 16 32 64 2>&1 | grep -v amdgpu
echo "== spmm bench rmat1m"
timeout 900 python scripts/spmm_bench.py 1000000 16 8 16 64 2>&1 | grep -v amdgpu
} 2>&1 | tee $OUT/log.txt
