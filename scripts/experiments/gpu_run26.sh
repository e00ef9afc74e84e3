#!/bin/bash
# SpGEMM v3c (LDS-class launches, staging reuse, shorter staging chain): parity + sweep + kernel trace
mkdir -p gpurun_out/r01z
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q -k "not config5" 2>&1 | tail -3
for cfg in "19 65536" "19 32768" "19 131072" "18 65536" "17 65536"; do set -- $cfg
  echo "--- winlog $1 heavy $2"
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_WINLOG=$1 SPGEMM_HEAVY=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 100 2>/dev/null | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "seconds|large_|small_" | cut -c1-66,110-200
done
echo "--- prof winlog 19 heavy 65536"
SPGEMM_PROF=1 SPGEMM_WINLOG=19 timeout 300 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof" | tail -2
} 2>&1 | tee gpurun_out/r01z/log_v3f.txt
