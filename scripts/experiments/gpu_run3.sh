#!/bin/bash
TAG=${1:-r01c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== x-load flavour A/B (rmat10m)"
for v in "--xload 0" "--xload 1" "--xload 2" "--xload 2 --tile 2048" "--xload 2 --idx-bytes 4"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['value'])"
done
echo "== laplace with xload 2"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --xload 2 --workload laplace4096 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['achieved'], d['value'])"
echo "== spgemm"
for cfg in "20000 8" "100000 8" "300000 8"; do timeout 300 python tests/spgemm_bench.py $cfg 2>&1 | tail -1; done
echo "== spgemm config 5 (1M, 8/row)"; timeout 600 python tests/spgemm_bench.py 1000000 8 2>&1 | tail -2
echo "== PMC rmat10m (xload 0)"; bash scripts/gpu_pmc.sh $TAG/pmc_rmat10m_x0 --xload 0
echo "== PMC rmat10m (xload 2)"; bash scripts/gpu_pmc.sh $TAG/pmc_rmat10m_x2 --xload 2
echo "== pytest spgemm durations"
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q --durations=8 2>&1 | tail -14
