#!/bin/bash
TAG=${1:-r01b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5
echo "== xmask experiments (timing only)"
for m in 1023 1048575; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --xmask $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('xmask $m', d['ms_per_step'], d['roofline']['achieved'])"
done
echo "== PMC"
bash scripts/gpu_pmc.sh $TAG/pmc_rmat10m
bash scripts/gpu_pmc.sh $TAG/pmc_laplace --workload laplace4096
