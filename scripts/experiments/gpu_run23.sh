#!/bin/bash
# SpGEMM v3: where does (winlog 19|18, heavy 65536) lose its time?  kernel trace + round counts
mkdir -p gpurun_out/r01z
export TMPDIR=/tmp
{
for cfg in "19 65536" "17 65536"; do set -- $cfg
  echo "--- prof winlog $1 heavy $2"
  SPGEMM_PROF=1 SPGEMM_WINLOG=$1 SPGEMM_HEAVY=$2 timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "spgemm_prof|seconds" | sed 's/"idx_bytes.*//'
  echo "--- kernel trace winlog $1 heavy $2"
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_WINLOG=$1 SPGEMM_HEAVY=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 100 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-200
done
} 2>&1 | tee gpurun_out/r01z/log_v3c.txt
