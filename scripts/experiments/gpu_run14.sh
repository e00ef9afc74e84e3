#!/bin/bash
TAG=${1:-r01n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest spgemm"
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== spgemm"
for cfg in "20000 8" "100000 8" "300000 8" "1000000 8"; do timeout 600 python tests/spgemm_bench.py $cfg 2>&1 | tail -1 | cut -c1-330; done
echo "== spgemm config 5 kernel stats"
( cd /tmp && rm -rf /tmp/sg && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/sg -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/sg -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-180 | head -8
} 2>&1 | tee $OUT/log.txt
