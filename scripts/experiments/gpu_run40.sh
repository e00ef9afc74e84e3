#!/bin/bash
mkdir -p gpurun_out/r01z
{
timeout 600 python -m pytest tests/test_spgemm_gpu.py tests/test_convert_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "--- A*A, 5-pt Laplacian 4096^2, entry-parallel small rows"
SPGEMM_MATRIX=laplace timeout 300 python tests/spgemm_bench.py 16777216 5 8 2000 2>&1 | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'
echo "--- config 5"
timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep seconds | sed 's/"nnz_a.*"seconds"/"seconds"/; s/"idx_bytes.*structure_bit/ structure_bit/'
} 2>&1 | tee gpurun_out/r01z/log_small_rows_v3_tiny.txt
