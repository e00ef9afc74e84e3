#!/bin/bash
TAG=${1:-r01s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spgemm"
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q --durations=4 2>&1 | tail -9
echo "== spgemm bucket table on/off, heavy sweep"
for b in 1 0; do for h in 65536; do
  for cfg in "300000 8" "1000000 8"; do echo -n "bucket=$b heavy=$h $cfg: "; timeout 600 python tests/spgemm_bench.py $cfg 8 300 $h $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['seconds'], d['gflops'], d['parity']['structure_bit_exact'], d['parity']['values_bit_exact'])"; done
done; done
echo "== config 5 kernel stats (defaults)"
( cd /tmp && rm -rf /tmp/sg && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/sg -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/sg -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-180 | head -8
} 2>&1 | tee $OUT/log.txt
