#!/bin/bash
mkdir -p gpurun_out/r11w
for v in no_parity no_spmv free_first ""; do
  echo "-- variant ${v:-as committed}" | tee -a gpurun_out/r11w/spmm_bench_bisect.jsonl
  SPMM_BENCH_VARIANT=$v timeout 200 python scripts/spmm_bench.py 10000000 32 8 16 2>&1 | grep -v amdgpu | cut -c1-140 | tee -a gpurun_out/r11w/spmm_bench_bisect.jsonl
done
