#!/bin/bash
# is the slow mode of the k = 16 SpMM call about the ADDRESSES of the hot rhs rows (hub columns at 0, 2^k, 2^j + 2^k)?  the same calls on a column-permuted matrix
mkdir -p gpurun_out/r11x
for ks in "16" "8 16"; do
  for p in "" 7; do
    echo "-- k = $ks, PERMUTE_COLS=${p:-off}" | tee -a gpurun_out/r11x/spmm_permuted_columns.jsonl
    PERMUTE_COLS=$p timeout 200 python scripts/spmm_bench.py 10000000 32 $ks 2>&1 | grep -v amdgpu | cut -c1-140 | tee -a gpurun_out/r11x/spmm_permuted_columns.jsonl
  done
done
