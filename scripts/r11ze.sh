#!/bin/bash
# SpMV R-MAT 10M on the banded plan: natural column ids against randomly permuted ones (bench.py --permute-cols)
mkdir -p gpurun_out/r11ze
for p in "" "--permute-cols 7" "--permute-cols 11"; do
  timeout 600 python bench.py --no-secondary --no-cpu-baseline $p 2>/dev/null | tee -a gpurun_out/r11ze/spmv_permuted_columns.jsonl | cut -c1-120
done
