#!/bin/bash
TAG=${1:-r01g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest all gpu"
timeout 1200 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14
echo "== default bench"
timeout 600 python bench.py 2>/dev/null | tee $OUT/bench_default.json | cut -c1-400
} 2>&1 | tee $OUT/log.txt
