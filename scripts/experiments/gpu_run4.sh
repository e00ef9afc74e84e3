#!/bin/bash
TAG=${1:-r01d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spmv"
timeout 900 python -m pytest tests/test_spmv_gpu.py -m gpu -x -q --durations=5 2>&1 | tail -15
echo "== XCS A/B (rmat10m)"
for v in "--xcs 2" "--xcs 1 --split 32" "--xcs 1 --split 64" "--xcs 1 --split 128" "--xcs 1 --split 512" "--xcs 0 --idx-bytes 4"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
echo "== other workloads (auto)"
for w in rmat1m laplace4096; do
  for v in "--xcs 2" "--xcs 0"; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload $w $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w $v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
  done
done
echo "== spgemm"
for cfg in "20000 8" "100000 8" "300000 8"; do timeout 300 python tests/spgemm_bench.py $cfg 2>&1 | tail -1; done
echo "== spgemm config 5 (1M, 8/row)"; timeout 900 python tests/spgemm_bench.py 1000000 8 2>&1 | tail -2
} 2>&1 | tee $OUT/log.txt
echo "== PMC rmat10m (auto = sliced)"; bash scripts/gpu_pmc.sh $TAG/pmc_rmat10m_xcs 2>&1 | tee -a $OUT/log.txt | grep -E "group|sliced_kernel|tile_kernel"
