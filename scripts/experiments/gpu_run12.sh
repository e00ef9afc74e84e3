#!/bin/bash
TAG=${1:-r01l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== occupancy cap A/B (LDS pad): 41 KB static -> 3 WGs/CU; pad 14000 -> 2 WGs; pad 45000 -> 1 WG"
for v in "" "--ldspad 14000" "--ldspad 45000" "--tile 2048 --ldspad 10000" "--xcs 2 --ldspad 14000" "--workload laplace4096 --ldspad 8000" "--workload laplace4096 --ldspad 16000"; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$v]', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
} 2>&1 | tee $OUT/log.txt
