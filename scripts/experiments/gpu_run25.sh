#!/bin/bash
# SpGEMM v3b (contiguous walk, branch-free search): parity + sweep + kernel trace
mkdir -p gpurun_out/r01z
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q -k "not config5" 2>&1 | tail -5
for cfg in "19 65536" "18 65536" "17 65536" "17 32768"; do set -- $cfg
  echo "--- winlog $1 heavy $2"
  SPGEMM_WINLOG=$1 SPGEMM_HEAVY=$2 timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep seconds | sed 's/"idx_bytes.*parity/ parity/'
done
for cfg in "19 65536" "17 65536"; do set -- $cfg
  echo "--- prof winlog $1 heavy $2"
  SPGEMM_PROF=1 SPGEMM_WINLOG=$1 SPGEMM_HEAVY=$2 timeout 300 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof" | tail -2
  echo "--- kernel trace winlog $1 heavy $2"
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_WINLOG=$1 SPGEMM_HEAVY=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "large_|small_" | cut -c1-60,110-200
done
} 2>&1 | tee gpurun_out/r01z/log_v3e.txt
