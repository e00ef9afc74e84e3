#!/bin/bash
# SpGEMM v3c A/B at winlog 17: dbg bits 1 = no staging reuse, 2 = conditional bucket loads, 4 = single launch
mkdir -p gpurun_out/r01z
export TMPDIR=/tmp
{
for dbg in 7 4 5 6 0; do
  echo "--- winlog 17 heavy 65536 dbg $dbg"
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_DBG=$dbg SPGEMM_WINLOG=17 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 10 2>/dev/null | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "seconds|large_" | cut -c1-66,110-200
done
} 2>&1 | tee gpurun_out/r01z/log_v3g.txt
