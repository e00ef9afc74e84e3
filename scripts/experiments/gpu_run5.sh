#!/bin/bash
TAG=${1:-r01e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spmv"
timeout 900 python -m pytest tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -4
echo "== XCS A/B (rmat10m)"
for v in "--xcs 2" "--xcs 1 --split 8" "--xcs 1 --split 32" "--xcs 1 --split 64" "--xcs 1 --split 256" "--xcs 1 --split 64 --idx-bytes 4"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
echo "== other workloads"
for w in rmat1m laplace4096; do
  for v in "--xcs 2" "--xcs 1"; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload $w $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w $v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
  done
done
} 2>&1 | tee $OUT/log.txt
echo "== kernel stats (auto)"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --xcs 1 > $GRAFT_REPO_ROOT/$OUT/stats_bench.json 2>/dev/null; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | cut -c1-190 | tee $OUT/kernel_stats.txt
