#!/bin/bash
mkdir -p gpurun_out/r01zv
export TMPDIR=/tmp
{
echo "# round 1 final build: SpGEMM config 5 (A*A, R-MAT 1M x 1M ~8/row, usize): wall time of the call, parity on 300 rows, kernel trace, phase profile"
timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep seconds
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-200
SPGEMM_PROF=1 timeout 300 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof" | tail -2
} 2>&1 | tee gpurun_out/r01zv/spgemm_final.txt
