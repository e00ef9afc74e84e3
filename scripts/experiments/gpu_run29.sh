#!/bin/bash
mkdir -p gpurun_out/r01z
{
for wl in 19 17; do
echo "--- prof winlog $wl"
SPGEMM_PROF=1 SPGEMM_WINLOG=$wl timeout 300 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof|seconds" | tail -3 | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'
done
} 2>&1 | tee gpurun_out/r01z/log_v4_expand_split.txt
