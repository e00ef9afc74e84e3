#!/bin/bash
mkdir -p gpurun_out/r01z
export TMPDIR=/tmp
{
echo "--- A*A, 5-pt Laplacian 4096^2 (all rows on the small-row wave/hash path)"
SPGEMM_MATRIX=laplace timeout 300 python tests/spgemm_bench.py 16777216 5 8 2000 2>&1 | grep seconds
( cd /tmp && rm -rf /tmp/st && SPGEMM_MATRIX=laplace timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 16777216 5 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-66,110-200 | head -8
} 2>&1 | tee gpurun_out/r01z/log_spgemm_laplace.txt
