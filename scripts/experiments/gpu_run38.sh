#!/bin/bash
mkdir -p gpurun_out/r01z
{
for mw in 11 12 14; do for hv in 32768 65536 131072; do
echo "--- minwin $mw heavy $hv"
SPGEMM_MINWIN=$mw SPGEMM_HEAVY=$hv timeout 300 python tests/spgemm_bench.py 1000000 8 8 20 2>&1 | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'
done; done
} 2>&1 | tee gpurun_out/r01z/log_v4_minwin.txt
