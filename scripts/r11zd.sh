#!/bin/bash
# SpGEMM config 5: does it matter that the hub rows / columns sit at 0, 2^k, 2^j + 2^k?  P A P^T with a random P against the control
mkdir -p gpurun_out/r11zd
for p in 0 7 11; do
  echo "-- SPGEMM_PERMUTE=$p" | tee -a gpurun_out/r11zd/spgemm_permuted.jsonl
  SPGEMM_PERMUTE=$p timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds" | cut -c1-330 | tee -a gpurun_out/r11zd/spgemm_permuted.jsonl
done
