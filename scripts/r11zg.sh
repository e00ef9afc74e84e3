#!/bin/bash
# SpGEMM with the bucket-table rows placed by a hash: gate, then natural labels against P A P^T
mkdir -p gpurun_out/r11zg
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
for p in "" 0 7; do
  echo "-- SPGEMM_PERMUTE=${p:-unset}" | tee -a gpurun_out/r11zg/spgemm_hashed_bucket_rows.jsonl
  SPGEMM_PERMUTE=$p timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds" | cut -c1-200 | tee -a gpurun_out/r11zg/spgemm_hashed_bucket_rows.jsonl
done
