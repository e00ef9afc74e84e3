#!/bin/bash
TAG=${1:-r01o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest virtual ranks"
timeout 600 python -m pytest tests/test_spmv_gpu.py -m gpu -x -q -k virtual 2>&1 | tail -3
echo "== virtual ranks scaling model"
timeout 900 python scripts/virtual_ranks.py 2>&1 | tail -1
} 2>&1 | tee $OUT/log.txt
