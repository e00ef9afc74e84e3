#!/bin/bash
mkdir -p gpurun_out/r11y
for p in 0 7 11; do
  echo "-- k = 16 alone, PERMUTE_COLS=$p (0 = identity: the control with the same allocation history)" | tee -a gpurun_out/r11y/spmm_permuted_columns_control.jsonl
  PERMUTE_COLS=$p timeout 200 python scripts/spmm_bench.py 10000000 32 16 2>&1 | grep -v amdgpu | cut -c1-140 | tee -a gpurun_out/r11y/spmm_permuted_columns_control.jsonl
done
