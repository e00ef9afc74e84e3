#!/bin/bash
mkdir -p gpurun_out/r01z
{
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q -k "not config5" 2>&1 | tail -2
for wl in 19 18 17 16; do
echo "--- winlog $wl"
SPGEMM_WINLOG=$wl timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'
done
for wl in 19 17; do
echo "--- prof winlog $wl"
SPGEMM_PROF=1 SPGEMM_WINLOG=$wl timeout 300 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof" | tail -2
done
} 2>&1 | tee gpurun_out/r01z/log_v4_interleaved_prefix.txt
