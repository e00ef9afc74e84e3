#!/bin/bash
# where should the automatic rule of spmm_relayout switch on?  smaller matrices, on against off
mkdir -p gpurun_out/r11zc
for cfg in "1000000 16" "1000000 64" "4000000 32" "200000 32"; do
  for o in "spmm_relayout=2" "spmm_relayout=1"; do
    echo "-- n nnz/row = $cfg, $o" | tee -a gpurun_out/r11zc/spmm_relayout_small.jsonl
    SPRS_OPTS="$o" timeout 200 python scripts/spmm_bench.py $cfg 8 16 32 2>&1 | grep -v amdgpu | cut -c1-120 | tee -a gpurun_out/r11zc/spmm_relayout_small.jsonl
  done
done
