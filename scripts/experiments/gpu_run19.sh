#!/bin/bash
TAG=${1:-r01v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spmv"
timeout 900 python -m pytest tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== sorted tiles A/B (rmat10m)"
for v in "--sort 0" "--sort 1" "--sort 1 --split 16" "--sort 1 --split 64" "--sort 1 --idx32 0" "--sort 1 --tile 2048"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$v]', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
echo "== rmat1m forced sliced + sort"
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload rmat1m --xcs 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['achieved'], d['value'])"
} 2>&1 | tee $OUT/log.txt
