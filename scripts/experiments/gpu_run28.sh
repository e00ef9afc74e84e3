#!/bin/bash
# SpGEMM v3 final defaults (winlog 17, single launch, result pool): full spgemm/convert tests + timings
mkdir -p gpurun_out/r01z
export TMPDIR=/tmp
{
timeout 1200 python -m pytest tests/test_spgemm_gpu.py tests/test_convert_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "--- defaults"
timeout 300 python tests/spgemm_bench.py 1000000 8 8 300 2>&1 | grep seconds
echo "--- u32 indices"
timeout 300 python tests/spgemm_bench.py 1000000 8 4 300 2>&1 | grep seconds
echo "--- kernel trace, defaults"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-66,110-200
echo "--- phase profile, defaults"
SPGEMM_PROF=1 timeout 300 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof" | tail -2
} 2>&1 | tee gpurun_out/r01z/log_v3_final.txt
