#!/bin/bash
# SpMM with the rhs gathered from a re-laid-out copy (option spmm_relayout): gate, then both allocation histories, on / off / auto
mkdir -p gpurun_out/r11za
timeout 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
for ks in "16" "8 16 32"; do
  for o in "spmm_relayout=2" "spmm_relayout=1" ""; do
    echo "-- k = $ks, ${o:-auto}" | tee -a gpurun_out/r11za/spmm_relayout_ab.jsonl
    SPRS_OPTS="$o" timeout 200 python scripts/spmm_bench.py 10000000 32 $ks 2>&1 | grep -v amdgpu | cut -c1-140 | tee -a gpurun_out/r11za/spmm_relayout_ab.jsonl
  done
done
