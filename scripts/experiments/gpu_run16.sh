#!/bin/bash
TAG=${1:-r01p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
for w in 0 5 10; do echo "== virtual ranks row_weight=$w"; ROW_WEIGHT=$w timeout 900 python scripts/virtual_ranks.py 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
for g in ('G=2','G=4','G=8'):
    e=d[g]; print(g, 'compute max/min', e['compute_ms_max'], e['compute_ms_min'], 'gather', e['modelled_allgather_ms'], 'step', e['modelled_step_ms'], 'GF', e['modelled_gflops'], 'rows', e['rows_per_block'])
"; done
} 2>&1 | tee $OUT/log.txt
