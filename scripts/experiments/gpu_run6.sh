#!/bin/bash
TAG=${1:-r01f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spmv"
timeout 900 python -m pytest tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -4
echo "== A/B (rmat10m)"
for v in "--xcs 1 --idx32 0" "--xcs 1 --idx32 1" "--xcs 1 --idx32 1 --tile 2048" "--xcs 1 --idx32 0 --tile 2048" "--xcs 2 --tile 2048" "--xcs 1 --idx32 1 --split 32" "--xcs 1 --idx32 1 --split 16" "--xcs 1 --idx-bytes 4"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
echo "== rmat1m forced / laplace"
for v in "--workload rmat1m --xcs 1" "--workload rmat1m --xcs 2" "--workload rmat1m --xcs 2 --tile 2048" "--workload laplace4096 --tile 2048"; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
} 2>&1 | tee $OUT/log.txt
