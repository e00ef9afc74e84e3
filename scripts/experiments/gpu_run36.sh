#!/bin/bash
mkdir -p gpurun_out/r01z
{
timeout 600 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q -k "not config5 and not window_sizes" 2>&1 | tail -2
for wl in 19 18 17 16; do
echo "--- 256-thread workgroups, winlog $wl"
SPGEMM_WINLOG=$wl timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'
done
} 2>&1 | tee gpurun_out/r01z/log_v4_block256.txt
